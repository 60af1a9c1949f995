#!/bin/bash
# registers, scratch and occupancy of every kernel of lart_amd/csrc as the compiler reports them (-Rpass-analysis=kernel-resource-usage):
#   tools/kernel_resources.sh [FILE.hip ...] > profiles/rNN_kernel_resources.txt
# A jump of ScratchSize or "VGPRs Spill" after a change is a performance bug (round 3: a run-time index into the kernel arguments copied the
# parameter block into scratch and made the peel march 2.5x slower).
cd "$(dirname "$0")/../lart_amd/csrc" || exit 1
FILES=${@:-transport_var0.hip transport_var1.hip transport_var2.hip transport_var3.hip transport_var4.hip transport_var7.hip transport_var9.hip transport_lt8.hip peel_kernels.hip}
for f in $FILES; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -mllvm -disable-machine-licm -mllvm -simplifycfg-sink-common=false -std=c++17 -fPIC -munsafe-fp-atomics \
     -Rpass-analysis=kernel-resource-usage -c $f -o /tmp/kr_$$.o 2>&1 | grep "remark:" | sed 's/.*remark: *//; s/ \[-Rpass.*//' | \
  awk -v f=$f '/Function Name/ {name=$3} /^VGPRs:/ {v=$2} /^ScratchSize/ {s=$3} /^Occupancy/ {o=$3} /^VGPRs Spill/ {sp=$3} /^LDS Size/ {printf "%-22s VGPRs %3d  scratch %5d B/lane  VGPR spills %3d  waves/SIMD %d  %s\n", f, v, s, sp, o, name}' | c++filt
done
rm -f /tmp/kr_$$.o
