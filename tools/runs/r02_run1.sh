cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
tools/micro/build/gemm_lab 8 > gpurun_out/lab1.log 2>&1
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace -d /tmp/prof_lab1 -- tools/micro/build/gemm_lab 2 > gpurun_out/lab1_prof.log 2>&1
python tools/rocpd_summary.py /tmp/prof_lab1/*/*.db 60 > gpurun_out/lab1_pmc.txt 2>&1
timeout 900 python -m pytest tests/test_bench_path_gpu.py -x -q > gpurun_out/t_new.log 2>&1
python tools/parity_report.py > gpurun_out/parity_report.json 2> gpurun_out/parity_report.err
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/t_all.log 2>&1
tail -5 gpurun_out/lab1.log gpurun_out/t_new.log gpurun_out/t_all.log
