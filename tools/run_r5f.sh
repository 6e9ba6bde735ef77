mkdir -p gpurun_out/r5f
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r5f/tests.log 2>&1; tail -5 gpurun_out/r5f/tests.log
timeout -k 10 400 python bench.py > gpurun_out/r5f/bench.json 2> gpurun_out/r5f/bench.err; tail -3 gpurun_out/r5f/bench.err; python - <<'PY'
import json
r=json.load(open('gpurun_out/r5f/bench.json'))
print(r['value'], r['ms_per_step'], r['roofline']['frac'])
s=r['secondary']
for k,v in s['small_batches']['batches'].items(): print('nq',k,v['ms_per_search'],v['phases_ms'],v['roofline']['achieved'])
print('k1001',s['k1001']['ms_per_step'],s['k1001']['phases_ms'])
print('msmarco',s['msmarco_scale']['ms_per_step'], s['msmarco_scale']['roofline']['frac'])
print('inbatch',{k:s['inbatch_b1024'][k] for k in ('hip_ms','kernels_ms','torch_fp32_ms')})
print('encode',s['encode_passages'].get('layer_kernels'),s['encode_passages'].get('torch_modules'))
b=s['bm25']; print('bm25',b.get('ms_per_call'),b.get('path'),b.get('cpu_baseline'),b.get('roofline',{}).get('frac'))
PY
