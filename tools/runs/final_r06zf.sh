# round-6 closing run: GPU suite, default bench (compact line + detail), rocprofv3 kernel stats of the bench
python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > gpurun_out/r06zf_gputest.log
python bench.py > gpurun_out/r06zf_bench.log 2>&1
tail -1 gpurun_out/r06zf_bench.log > gpurun_out/r06zf_bench_compact_line.json
cp gpurun_out/bench_detail.json gpurun_out/r06zf_bench_detail.json
cat gpurun_out/r06zf_gputest.log
python -c "
import json
d=json.load(open('gpurun_out/r06zf_bench_detail.json'))
print(d['ms_per_step'], d['roofline']['frac'], {k:(v.get('ms_per_step'), v.get('device_allocs_in_timed_region')) for k,v in d.items() if isinstance(v,dict) and 'ms_per_step' in v})
"
wc -c gpurun_out/r06zf_bench_compact_line.json
