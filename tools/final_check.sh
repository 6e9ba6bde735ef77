#!/bin/bash
# the round's closing run on the GPU box: the whole -m gpu suite, smoke(), the default bench line.  bash tools/final_check.sh
mkdir -p gpurun_out/final
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/final/tests.log 2>&1; tail -3 gpurun_out/final/tests.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout -k 10 400 python bench.py > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err; tail -2 gpurun_out/final/bench.err | cut -c1-200; python - <<'PY'
import json
r=json.load(open('gpurun_out/final/bench.json'))
print(r['value'], r['ms_per_step'], r['roofline']['frac'], r['roofline']['traffic'])
s=r['secondary']
for k,v in s['small_batches']['batches'].items(): print('nq',k,v['ms_per_search'],v['phases_ms']['search_total'],v['roofline']['achieved'])
print('k1001',s['k1001']['value'],s['k1001']['ms_per_step'])
print('msmarco',s['msmarco_scale']['value'],s['msmarco_scale']['ms_per_step'], s['msmarco_scale']['roofline']['frac'], s['msmarco_scale'].get('gpu_over_cpu'))
print('inbatch',{k:s['inbatch_b1024'][k] for k in ('hip_ms','kernels_ms','torch_fp32_ms','torch_bf16_autocast_ms')}, s['inbatch_b1024']['roofline']['frac'])
e=s['encode_passages']; print('encode',e.get('layer_kernels',{}).get('value'),e.get('torch_modules',{}).get('value'),e.get('roofline',{}).get('frac'))
b=s['bm25']; print('bm25',b.get('value'),b.get('ms_per_call'),b.get('cpu_baseline',{}).get('recall_of_gpu_ids'),b.get('roofline',{}).get('traffic'),b.get('gpu_over_cpu'))
print('cpu',r['cpu_baseline']['value'],r['cpu_baseline']['cores'],r['cpu_baseline']['recall_at_k_of_gpu_vs_cpu'])
PY
