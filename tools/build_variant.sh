# usage: bash tools/build_variant.sh <name> [-DFLAG ...]  -> rainbow_amd/librainbow_hip_<name>.so (same sources, extra -D switches)
# for same-box A/B runs: RAINBOW_AMD_LIB=$PWD/rainbow_amd/librainbow_hip_<name>.so python bench.py ...
set -e
NAME=$1; shift
CS=rainbow_amd/csrc
OBJS=""
HASH=$(python -c "import __graft_entry__ as g; print(g.source_hash())")     # bench.py refuses a library without its source hash
for f in $CS/*.hip; do
  o=/tmp/rbv_${NAME}_$(basename $f .hip).o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-result -DRB_SOURCE_HASH=\"$HASH\" "$@" -c $f -o $o &
  OBJS="$OBJS $o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o rainbow_amd/librainbow_hip_${NAME}.so
echo built rainbow_amd/librainbow_hip_${NAME}.so
