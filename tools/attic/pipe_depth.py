import sys, time
sys.path.insert(0,'/root/repo')
import torch
from mesh2splat_amd import synth
from mesh2splat_amd.converter import Converter
scene = synth.cube_sphere(289, tex_size=2048)
c = Converter(0); c.upload_scene(scene)
R=1024
for _ in range(5): c.convert(R)
def run(depth, prof, K=200):
    c.set_profiling(prof)
    torch.cuda.synchronize()
    t0=time.perf_counter()
    inflight=0
    for i in range(K):
        c.submit(R); inflight+=1
        if inflight>=depth: c.wait(); inflight-=1
    while inflight: c.wait(); inflight-=1
    torch.cuda.synchronize()
    return (time.perf_counter()-t0)/K*1e3
for prof in (False, True):
    for depth in (1,2,3,4):
        print('prof',prof,'depth',depth, round(run(depth,prof),4),'ms/step')
c.set_profiling(False)
t0=time.perf_counter()
for i in range(200): c.convert(R)
print('sync convert', round((time.perf_counter()-t0)/200*1e3,4))
