#!/bin/bash
# round 5: everything profiles/r05_* holds, in one gpurun call.
#   tools/r05/profile_round5.sh  ->  gpurun_out/r05/..., gpurun_out/e2e_pass/..., gpurun_out/x3_pmc/..., gpurun_out/r05_roi/...
set -u
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
bash tools/profile_round.sh r05 > gpurun_out/r05_profile_round.log 2>&1
# the float32 split-precision pass, kernel by kernel (15 images), next to the exact-float32 pass
DT=fp32 BATCH=15 EXTRA="--f32-form x3" bash tools/exp/e2e_pass_trace.sh > /dev/null 2>&1; cp gpurun_out/e2e_pass/pass.txt gpurun_out/r05/e2e_fp32_x3_b15_pass_trace.txt
DT=fp32 BATCH=15 EXTRA="--f32-form exact" bash tools/exp/e2e_pass_trace.sh > /dev/null 2>&1; cp gpurun_out/e2e_pass/pass.txt gpurun_out/r05/e2e_fp32_exact_b15_pass_trace.txt
DT=fp16 BATCH=1 EXTRA="" bash tools/exp/e2e_pass_trace.sh > /dev/null 2>&1; cp gpurun_out/e2e_pass/pass.txt gpurun_out/r05/e2e_fp16_b1_pass_trace.txt
# its layers and its counters on the RpnHead's P2 level
python3 tools/r05/x3_layers.py --batch 15 --out gpurun_out/r05/x3_layers_b15.json > gpurun_out/r05/x3_layers_b15.txt 2>&1
python3 tools/r05/x3_layers.py --batch 1 --out gpurun_out/r05/x3_layers_b1.json > gpurun_out/r05/x3_layers_b1.txt 2>&1
bash tools/r05/x3_pmc.sh 15,200,334,256,512,3 x3 > gpurun_out/r05/x3_pmc_rpn_p2.txt 2>&1
# the two-limb float16 form: its pass, its counters, its tiles, what the parts of the loop cost (diagnostic libraries: build them
# first in the build container, tools/r05/x2_diag.sh build), the own pointwise kernel against the library
DT=fp32 BATCH=15 EXTRA="--f32-form x2" bash tools/exp/e2e_pass_trace.sh > /dev/null 2>&1; cp gpurun_out/e2e_pass/pass.txt gpurun_out/r05/e2e_fp32_x2_b15_pass_trace.txt
DT=fp32 BATCH=1 EXTRA="--f32-form x2" bash tools/exp/e2e_pass_trace.sh > /dev/null 2>&1; cp gpurun_out/e2e_pass/pass.txt gpurun_out/r05/e2e_fp32_x2_b1_pass_trace.txt
bash tools/r05/x3_pmc.sh 15,200,334,256,512,3 x2 > gpurun_out/r05/x2_pmc_rpn_p2.txt 2>&1
python3 tools/r05/x3_tiles.py 15 1 x2 > gpurun_out/r05/x2_tiles_b15.txt 2>&1
python3 tools/r05/x3_layers.py --check > gpurun_out/r05/x2_layers_vs_float64.txt 2>&1
python3 tools/r05/e2e_x3.py x2:30,x3:30,x2:1,x3:1 > gpurun_out/r05/x2_e2e.txt 2>&1
ls tools/exp/libodet_x3_ALL3.so > /dev/null 2>&1 && bash tools/r05/x2_diag.sh > gpurun_out/r05/x2_x3_loop_parts.txt 2>&1
for b in 1 4 30; do python3 tools/exp/pointwise_layers.py $b; done > gpurun_out/r05/pointwise_vs_library.txt 2>&1
# the single-level RoI forms (times) and the HBM-side bytes of the tensorpack RoIAlign layer with one slice per XCD
python3 tools/roi_forms.py > gpurun_out/r05/roi_forms_times.json 2> gpurun_out/r05/roi_forms_times.err
FORMS="6" bash tools/pmc_roi_forms.sh gpurun_out/r05_roi > gpurun_out/r05/roi_forms_pmc.log 2>&1
cp gpurun_out/r05_roi/roi_forms_pmc.json gpurun_out/r05/roi_forms_pmc_form6.json 2>/dev/null
ls gpurun_out/r05 | head -40
