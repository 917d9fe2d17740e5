#!/bin/bash
# usage: tools/spill_report.sh [out-file]  -> every kernel of the library with its VGPR count and scratch bytes per lane (hipcc
# -Rpass-analysis=kernel-resource-usage, the flags of tmg_hip.build); kernels with scratch > 0 are listed first.  CPU only (cross-compiles).
cd "$(dirname "$0")/.."
OUT=${1:-/dev/stdout}
TMP=$(mktemp -d)
# (the matrix-core files are built without the packed-fp32 instructions: tmg_hip.NO_PACKED_F32)
NOPK=$(python3 -c "import sys; sys.path.insert(0, 'deep-turbulence_amd'); import tmg_hip; print(' '.join(tmg_hip.NO_PACKED_F32))")
for f in tmg_conv tmg_pointwise tmg_physics tmg_mix16 tmg_coupling tmg_wino tmg_thin tmg_glue; do
  EXTRA=""
  case " $NOPK " in *" $f.hip "*) EXTRA="-Xclang -target-feature -Xclang -packed-fp32-ops";; esac
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result $EXTRA -I include -c deep-turbulence_amd/csrc/$f.hip -o $TMP/$f.o \
    -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "Function Name|VGPRs:|ScratchSize" | paste - - - \
    | sed 's/\[-Rpass-analysis=kernel-resource-usage\]//g' \
    | awk -v f=$f '{n=""; v=""; s=""; for(i=1;i<=NF;i++){ if($i=="Name:")n=$(i+1); if($i=="VGPRs:")v=$(i+1); if($i=="[bytes/lane]:")s=$(i+1);} printf "%-14s VGPRs %4s  scratch %5s B/lane  %s\n", f, v, s, n}' > $TMP/$f.txt &
done
wait
{ echo "# kernels with scratch (spilled registers):"; cat $TMP/*.txt | awk '$5+0 > 0' | c++filt; echo "# all kernels:"; cat $TMP/*.txt | c++filt; } > "$OUT"
rm -rf $TMP
