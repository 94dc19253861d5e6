"""Micro-benchmarks of the HBM-bound kernels against the 8 TB/s roof.
Prints a markdown table: algorithmic bytes per call / measured time.  Run on the GPU box."""
import sys; sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import torch, numpy as np
from selfreconcode_amd.ext import FastMinv, GridSamplerMine, MCGpu, interp2x_boundary3d
from selfreconcode_amd.synthetic import synthetic_lbs_volume, synthetic_joints, LBS_BMIN, LBS_BMAX, SMPL_PARENTS
from selfreconcode_amd.model.Deformer import LBSkinner
from selfreconcode_amd.utils import smpl_tmp_Apose
dev='cuda:0'
def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize(); s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e)/n
rows=[]
N=4_000_000
m=torch.randn(N,3,3,device=dev)
ms=timeit(lambda: FastMinv.Fast3x3Minv(m)); rows.append(('minv fwd', N, 73*N, ms))
inv,_=FastMinv.Fast3x3Minv(m); g=torch.randn_like(m)
ms=timeit(lambda: FastMinv.Fast3x3Minv_backward(g,inv)); rows.append(('minv bwd', N, 108*N, ms))
vol=synthetic_lbs_volume((65,225,129), device=dev).contiguous(memory_format=torch.channels_last_3d)
P=1_000_000
grid=(torch.rand(1,1,1,P,3,device=dev)*2-1)
ms=timeit(lambda: GridSamplerMine.forward(vol,grid,0,1)); rows.append(('gridsample fwd (channel-last vol, 24 ch)', P, 796*P, ms))
volc=vol.contiguous()
ms=timeit(lambda: GridSamplerMine.forward(volc,grid,0,1)); rows.append(('gridsample fwd (reference NCDHW layout)', P, 796*P, ms))
go=torch.randn(1,24,1,1,P,device=dev)
ms=timeit(lambda: GridSamplerMine.backward(vol,grid,go,0,1,want_grad_input=False)); rows.append(('gridsample bwd (grad_grid only)', P, (796+96)*P, ms))
skin=LBSkinner(vol, LBS_BMIN, LBS_BMAX, synthetic_joints(), np.array(SMPL_PARENTS), init_pose=torch.from_numpy(smpl_tmp_Apose(1))).to(dev)
poses=torch.randn(3,24,3,device=dev)*0.1; trans=torch.zeros(3,3,device=dev)
A=skin.posed_transforms(poses)
pts=((torch.rand(P,3,device=dev)-0.5)*torch.tensor([1.4,2.0,0.7],device=dev)); bi=torch.randint(0,3,(P,),device=dev)
ms=timeit(lambda: skin.fused(pts,A,trans,bi,False)); rows.append(('lbs fused fwd', P, 796*P, ms))
ms=timeit(lambda: skin.fused(pts,A,trans,bi,True)); rows.append(('lbs fused fwd + jacobian', P, (796+36)*P, ms))
yb=torch.randn(P,3,device=dev)
ms=timeit(lambda: skin.fused_backward(pts,A,bi,0,yb,True,True,True)); rows.append(('lbs fused bwd (random frame per point)', P, (796+24)*P, ms))
bs=torch.sort(bi).values
ms=timeit(lambda: skin.fused_backward(pts,A,bs,0,yb,True,True,True)); rows.append(('lbs fused bwd (points grouped by frame)', P, (796+24)*P, ms))
for n in (257,513):
    x,y,z=torch.meshgrid(*[torch.linspace(-1,1,n,device=dev)]*3, indexing='ij')
    sdf=(torch.sqrt(x*x+0.8*y*y+z*z)-0.63+0.01*torch.sin(20*x)).contiguous()
    ms=timeit(lambda: MCGpu.mc_gpu(sdf,1.,1.,1.,0.,0.,0.,0.), n=5)
    v,f=MCGpu.mc_gpu(sdf,1.,1.,1.,0.,0.,0.,0.)
    rows.append((f'marching cubes {n}^3 (V={v.shape[0]}, F={f.shape[0]})', n**3, 4*n**3+12*v.shape[0]+24*f.shape[0], ms))
c=torch.randn(1,1,257,257,257,device=dev)
ms=timeit(lambda: interp2x_boundary3d.forward(c,0.0), n=5); rows.append(('interp2x_boundary3d fwd 257^3 -> 513^3', 513**3, 4*257**3+5*513**3, ms))
# round 2: LBS value+Jacobian backward, the two rasterisers at the bench's sizes (3 x 86k template vertices / 172k faces, 540^2)
from selfreconcode_amd.model.Deformer import _LBSValueJacobian
Pj=200_000
qj=pts[:Pj].clone().requires_grad_(True); bj=torch.sort(bi[:Pj]).values
yj,Jj=_LBSValueJacobian.apply(skin,qj,A.detach(),trans,bj,0)
wy=torch.randn_like(yj); wJ=torch.randn_like(Jj)
ms=timeit(lambda: torch.autograd.grad([yj,Jj],[qj],[wy,wJ],retain_graph=True)); rows.append(('lbs value+jacobian bwd (grouped by frame)', Pj, (2*796+60)*Pj, ms))
from selfreconcode_amd.ops import points_silhouette, rasterize_meshes
from selfreconcode_amd.model.CameraMine import RectifiedPerspectiveCameras
H=W=540
cam=RectifiedPerspectiveCameras(torch.tensor([[648.,648.]],device=dev),torch.tensor([[270.,270.]],device=dev),torch.diag(torch.tensor([-1.,1.,-1.],device=dev))[None],torch.tensor([[0.,0.15,2.4]],device=dev),[(W,H)])
n=257
x,y,z=torch.meshgrid(*[torch.linspace(-0.8,0.8,n,device=dev)]*3, indexing='ij')
vs,fs=MCGpu.mc_gpu((torch.sqrt(x*x+0.7*y*y+z*z)-0.6).contiguous(),1.6/256,1.6/256,1.6/256,-0.8,-0.8,-0.8,0.)
vs=vs[None].expand(3,-1,3).contiguous()+torch.randn(3,1,3,device=dev)*0.01
xy,zz=cam.project_ndc(vs)
ms=timeit(lambda: points_silhouette(xy,zz,H,W,0.006,50)); rows.append((f'point silhouette fwd K=50 (3 x {vs.shape[1]} points, 540^2)', 3*vs.shape[1], 12*3*vs.shape[1]+4*3*H*W, ms))
xyg=xy.clone().requires_grad_(True); mk=points_silhouette(xyg,zz,H,W,0.006,50); gm=torch.randn_like(mk)
ms=timeit(lambda: torch.autograd.grad(mk,xyg,gm,retain_graph=True)); rows.append(('point silhouette bwd', 3*vs.shape[1], 20*3*vs.shape[1]+16*3*H*W, ms))
ms=timeit(lambda: rasterize_meshes(xy,zz,fs,H,W)); rows.append((f'mesh rasteriser (3 x {fs.shape[0]} faces, 540^2)', 3*fs.shape[0], 3*fs.shape[0]*(24+36)+3*H*W*(8+8+12+4), ms))
print("| kernel | units | algorithmic bytes | ms | GB/s | % of 8 TB/s |"); print("|---|---|---|---|---|---|")
for name,u,b,ms in rows:
    print(f"| {name} | {u} | {b/1e6:.1f} MB | {ms:.3f} | {b/ms/1e6:.0f} | {b/ms/1e6/8000*100:.1f} |")
