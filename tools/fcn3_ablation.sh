#!/bin/bash
# Same-box ablation of the FourCastNet3 step (tools/fcn3_step.py): every row switches ONE more of this round's changes on.
#   bash tools/fcn3_ablation.sh > gpurun_out/fcn3_ablation.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
run() {
  echo "== $1"
  shift
  env "$@" timeout 400 python $R/tools/fcn3_step.py 2 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
f = d['hip_kernel_families_ms']
g = lambda key: sum(v['ms_total'] for k, v in f.items() if key in k)
print(f\"   step {d['ms_per_step']:8.1f} ms   peak {d['peak_hbm_gb']:6.1f} GB   DISCO fwd {g('disco_fwd'):7.1f}  bwd {g('disco_bwd'):7.1f}   channel GEMM nn {g('conv1x1_nn'):6.1f}  wgrad {g('conv1x1_wgrad'):6.1f}   resample {g('resample'):5.1f}   (ms per step, HIP events)\")
"
}
run "list kernels (csrc/disco.hip), round-1 tile GEMMs for FCN3's channel counts" MAKANI_AMD_DISCO=lists MAKANI_AMD_CONV_RINGK=0 MAKANI_AMD_WGRAD_2D=0
run "+ run-form kernels, one stream per basis function (forward and adjoint)" MAKANI_AMD_DISCO_FUSED=0 MAKANI_AMD_DISCO_ADJ=lists MAKANI_AMD_DISCO_MIXFIRST=0 MAKANI_AMD_CONV_RINGK=0 MAKANI_AMD_WGRAD_2D=0
run "+ forward for all nine basis functions per stream" MAKANI_AMD_DISCO_ADJ=lists MAKANI_AMD_DISCO_MIXFIRST=0 MAKANI_AMD_CONV_RINGK=0 MAKANI_AMD_WGRAD_2D=0
run "+ ring GEMM kernels for any channel count (forward / data gradient, 2-D weight gradient)" MAKANI_AMD_DISCO_ADJ=lists MAKANI_AMD_DISCO_MIXFIRST=0
run "+ channel mix first where there are fewer output than input channels (decoders)" MAKANI_AMD_DISCO_ADJ=lists
run "+ data gradient through the transposed one-in-K-out kernel (everything on: the default)" MAKANI_AMD_DISCO=runs
