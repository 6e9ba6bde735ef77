mkdir -p gpurun_out/dbg
C="--steps 3 --warmup 1 --rows 180000 --queries 200 --k 1001 --data sorted --cpu-queries 0 --no-secondary"
python bench.py --gpus 1 --dump-ids gpurun_out/dbg/one.pt $C > gpurun_out/dbg/one.json 2> gpurun_out/dbg/one.err
python bench.py --gpus 3 --dist-backend gloo --same-device --dump-ids gpurun_out/dbg/three.pt $C > gpurun_out/dbg/three.json 2> gpurun_out/dbg/three.err
CCREC_SHORT_LISTS=0 python bench.py --gpus 3 --dist-backend gloo --same-device --dump-ids gpurun_out/dbg/three_full.pt $C > gpurun_out/dbg/three_full.json 2> gpurun_out/dbg/three_full.err
python - <<'PY'
import torch, json
a=torch.load('gpurun_out/dbg/one.pt'); b=torch.load('gpurun_out/dbg/three.pt'); c=torch.load('gpurun_out/dbg/three_full.pt')
for name,x in (('short',b),('full',c)):
    d=(a!=x)
    rows=d.any(1).nonzero().squeeze(1)
    print(name,'rows differing',len(rows),'entries',int(d.sum()))
    for r in rows[:5].tolist():
        pos=d[r].nonzero().squeeze(1)
        print('  row',r,'first pos',pos[:6].tolist(),'n',len(pos),'set equal',set(a[r].tolist())==set(x[r].tolist()), 'a',a[r][pos[:4]].tolist(),'x',x[r][pos[:4]].tolist())
print(json.load(open('gpurun_out/dbg/three.json'))['exchange'])
PY
