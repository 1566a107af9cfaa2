#!/bin/bash
# VGPR / SGPR / LDS / scratch per kernel of a built libags_raster.so (default: the in-tree one); works without a GPU.
# usage: profiles/experiments/kernel_regs.sh [path/to/lib.so] [grep pattern]
set -e
LIB=$(readlink -f "${1:-active-gs_amd/lib/libags_raster.so}")
PAT="${2:-.}"
T=$(mktemp -d); cp "$LIB" "$T/lib.so"; cd "$T"
/opt/rocm/lib/llvm/bin/llvm-objdump --offloading lib.so > /dev/null
for f in lib.so.*gfx950; do
  /opt/rocm/lib/llvm/bin/llvm-readelf --notes "$f" | grep -E "^\s+\.name:|\.vgpr_count|\.sgpr_count|\.private_segment_fixed_size|\.group_segment_fixed_size|vgpr_spill"
done | awk '/\.name:/{name=$2} /group_segment/{g=$2} /private_segment/{p=$2} /sgpr_count/{s=$2} /vgpr_count/{v=$2} /vgpr_spill/{print "vgpr",v,"sgpr",s,"lds",g,"scratch",p,"spill",$2, name}' | grep -E "$PAT" | cut -c1-170
rm -rf "$T"
