#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-r05f}; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_bf16_gpu.py tests/test_model_gpu.py -m gpu -q -s -k "any_patch or autocast_parameter or short_training or inference_full_bag" > $OUT/pytest_new.log 2>&1; echo "new tests rc $?"; grep -E "passed|failed|HIP bf16|loss:|max relative|Error|assert" $OUT/pytest_new.log | head -120
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $OUT/pytest.log
