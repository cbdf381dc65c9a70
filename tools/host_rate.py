import time, torch, sys
sys.path.insert(0, '/root/repo')
import tetris_piclim as T
n=4096
env=T.BatchedTetris(10,40,n,auto_reset=True)
rows,pieces=env.synthetic_configs(n); env.load_configs(rows,pieces); env.reset()
a=env.synthetic_actions(0); r=torch.empty(n,dtype=torch.float32,device=env.device); d=torch.empty(n,dtype=torch.uint8,device=env.device)
for _ in range(200): env.step_into(a,r,d)
torch.cuda.synchronize()
K=5000
t0=time.perf_counter()
for _ in range(K): env.step_into(a,r,d)
t1=time.perf_counter()
torch.cuda.synchronize()
t2=time.perf_counter()
print(f"host enqueue {1e6*(t1-t0)/K:.2f} us/step; with drain {1e6*(t2-t0)/K:.2f} us/step")
