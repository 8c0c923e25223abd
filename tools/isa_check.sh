# usage: bash tools/isa_check.sh — gfx950 ISA of every translation unit: kernels, and any FLAT or SCRATCH instruction (there must be none:
# a kernel with a scratch segment slows every kernel of the stream; flat loads probe the LDS and scratch apertures)
for f in learner replay common; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off --cuda-device-only -S rainbow_amd/csrc/$f.hip -o /tmp/isa_$f.s 2>/dev/null
  echo "$f: $(grep -c '^\s*.amdhsa_kernel ' /tmp/isa_$f.s) kernels"
  awk '/^[_A-Za-z0-9]+:/{name=$1} /scratch_load|scratch_store|flat_load|flat_store|flat_atomic/{if ($0 !~ /^\s*;/) print "   OFFENDER", name, $1}' /tmp/isa_$f.s | sort | uniq -c
  grep -A40 "^\s*.amdhsa_kernel " /tmp/isa_$f.s | awk '/.amdhsa_kernel /{k=$2} /.amdhsa_private_segment_fixed_size/{if ($2 != 0) print "   SCRATCH SEGMENT", k, $2}'
done
