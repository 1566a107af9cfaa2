# same-box A/B of library builds on the large configurations: bash profiles/experiments/ab_large.sh <runs> tagA tagB ...
# (tag "cur" = the in-tree library, others scratch/libags_<tag>.so)
runs=$1; shift
for r in $(seq 1 $runs); do for tag in "$@"; do
  if [ "$tag" = cur ]; then unset AGS_LIB_PATH; else export AGS_LIB_PATH=$PWD/scratch/libags_$tag.so; fi
  python examples/large_configs.py 2>&1 | tail -2 | python3 -c "import sys,json; [print('$tag', json.loads(l)['config'][:3], json.loads(l)['ms_per_step']) for l in sys.stdin if l.startswith('{')]"
done; done
unset AGS_LIB_PATH
