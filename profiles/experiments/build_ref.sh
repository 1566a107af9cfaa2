# build the library of another commit into scratch/libags_<tag>.so (same-box A/B against the working tree):
#   bash profiles/experiments/build_ref.sh <commit> <tag>
set -e
commit=$1; tag=$2; root=$(git rev-parse --show-toplevel)
rm -rf /tmp/ags_ref_$tag; git worktree add -f /tmp/ags_ref_$tag $commit > /dev/null 2>&1
(cd /tmp/ags_ref_$tag && python -c "
import sys; sys.path.insert(0, '.')
import active_gs_amd.build as b; print(b.build(force=True))")
mkdir -p $root/scratch; cp /tmp/ags_ref_$tag/active-gs_amd/lib/libags_raster.so $root/scratch/libags_$tag.so
git worktree remove --force /tmp/ags_ref_$tag; echo scratch/libags_$tag.so
