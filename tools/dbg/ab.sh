python tools/dbg/qa_debug.py 2>&1 | tail -16 | grep -v "first\|per seg"
for i in 1 2; do for v in rot rot2; do echo "== $v"; RC_HIP_LIB=realcamnet_amd/_alt/lib_$v.so python tools/gma_stage_bench.py 2>&1 | grep "ONE launch\|whole block"; done; done
python -m pytest tests/test_gma.py -x -q -m gpu 2>&1 | tail -3
