cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for env in "" "HSA_ENABLE_INTERRUPT=0"; do
for i in 1 2; do
env $env python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-pipelined 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())
print('[$env] driver-like: step %.1f us value %.3f M kernel %.1f us' % (d['ms_per_step']*1e3, d['value']/1e6, d['roofline']['kernel_ms']*1e3))"
done
done
python3 - <<'PY'
import time, torch, sys
sys.path.insert(0,'.')
x = torch.zeros(1, device='cuda')
for _ in range(100): x += 1
torch.cuda.synchronize()
ts=[]
for _ in range(20):
    t0=time.perf_counter(); torch.cuda.synchronize(); ts.append(time.perf_counter()-t0)
print('empty synchronize: %.1f us' % (1e6*sum(ts)/len(ts)))
ts=[]
for _ in range(20):
    torch.cuda.synchronize(); t0=time.perf_counter(); x += 1; torch.cuda.synchronize(); ts.append(time.perf_counter()-t0)
print('one tiny kernel + synchronize: %.1f us' % (1e6*sum(ts)/len(ts)))
PY
