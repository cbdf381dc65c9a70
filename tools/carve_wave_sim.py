import numpy as np, heapq, sys, os
# search lengths of 20,000 configurations at L = 10, M = 40 without the restart rule (what iteration_histogram.json summarises)
its=np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'profiles', 'r04_carve', 'search_lengths_L10_M40.npz'))['iterations'].astype(np.int64)
rng=np.random.default_rng(1)
def sample(): return int(its[rng.integers(len(its))])
def run_wave(C, T0, policy, H=8, theta=0.0, A=24):
    # returns tail length (trips after T0) and total lane-iterations in the tail
    # main phase: each lane runs configs sequentially (attempt chain) until time >= T0 when it takes no new config
    t=0
    lanes=[]  # per lane: busy_until, and what
    # config records
    cfgs=[]   # dict: attempts {a:(start,end,success)}, ticket, resolved_time
    events=[] # (time, lane)
    lane_job=[None]*64
    def start(lane,c,a,now):
        x=sample(); lim=C<<(a//6)
        ok = x<=lim; dur = x if ok else lim
        cfgs[c]['att'][a]=[now,now+dur,ok,lane,False]  # start,end,ok,lane,dropped
        lane_job[lane]=(c,a)
        heapq.heappush(events,(now+dur,lane,c,a))
    def new_cfg(lane,now):
        cfgs.append({'att':{}, 'ticket':0,'done':None,'home':lane})
        start(lane,len(cfgs)-1,0,now)
    for l in range(64): new_cfg(l,0)
    idle=set()
    def resolved(c):
        return cfgs[c]['done'] is not None
    def try_resolve(c,now):
        cf=cfgs[c]
        fin=[a for a,v in cf['att'].items() if v[2] and v[1]<=now and not v[4]]
        if not fin: return
        w=min(fin)
        if all((b in cf['att'] and cf['att'][b][1]<=now and not cf['att'][b][2]) for b in range(w)):
            cf['done']=now
    def helpers(now):
        # assign idle lanes
        if not idle: return
        open_c=[c for c in range(len(cfgs)) if not resolved(c)]
        cands=[]
        for c in open_c:
            cf=cfgs[c]
            fin=any(v[2] and v[1]<=now for v in cf['att'].values())
            if fin: continue
            inflight=sum(1 for v in cf['att'].values() if v[1]>now and not v[4])
            if inflight>=H or cf['ticket']+1>=A: continue
            # age of lowest running attempt relative to its limit
            low=min((a for a,v in cf['att'].items() if v[1]>now and not v[4]),default=None)
            if low is None: continue
            v=cf['att'][low]; lim=C<<(low//6); age=(now-v[0])/lim
            if age<theta: continue
            cands.append((-(age) if policy=='age' else c, c))
        cands.sort()
        for (_,c) in cands:
            if not idle: break
            l=idle.pop()
            cfgs[c]['ticket']+=1
            start(l,c,cfgs[c]['ticket'],now)
    lane_iters_tail=0
    while events:
        now,lane,c,a=heapq.heappop(events)
        v=cfgs[c]['att'][a]
        if v[4] or lane_job[lane]!=(c,a): continue
        lane_job[lane]=None
        cf=cfgs[c]
        if resolved(c):
            pass
        else:
            try_resolve(c,now)
        # drop attempts above a finished one
        if v[2]:
            for b,u in cf['att'].items():
                if b>a and u[1]>now and not u[4]:
                    u[4]=True; lane_job[u[3]]=None; idle.add(u[3])
        # what does this lane do next
        if not resolved(c) and not v[2] and cf['home']==lane and not any(u[2] and u[1]<=now for u in cf['att'].values()) and cf['ticket']+1<A:
            cf['ticket']+=1; start(lane,c,cf['ticket'],now); 
        elif now<T0 and cf['home']==lane and resolved(c):
            new_cfg(lane,now)
        elif now<T0 and cf['home']==lane and not resolved(c):
            idle.add(lane)   # owner waiting (rare before T0)
        else:
            idle.add(lane)
        if now>=T0: helpers(now)
        # check all resolved & now>=T0
    end=max(cf['done'] for cf in cfgs)
    return end-T0, len(cfgs)
for C,policy,H,theta in ((3328,'lane',8,0),(3328,'age',8,0),(3328,'age',8,0.5),(3328,'age',4,0.5),(3328,'age',2,0.0),(3328,'age',3,0.3),(2048,'age',8,0.3),(1024,'age',8,0.0),(1024,'lane',8,0),(100000,'lane',8,0)):
    r=[run_wave(C,7400,policy,H,theta) for _ in range(60)]
    tails=np.array([x[0] for x in r])
    print(C,policy,H,theta,'tail trips median',np.median(tails),'p90',np.percentile(tails,90),'max',tails.max(),'configs/wave',np.mean([x[1] for x in r]))
