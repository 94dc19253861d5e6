"""Per-shape GEMM time inside real training iterations (HIP events around every launch): where the layer-GEMM time goes."""
import sys, time, collections; sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import torch
from selfreconcode_amd.synthetic import build_synthetic_scene
from selfreconcode_amd import mlp_engine as me
net, ds, conf = build_synthetic_scene()
me.set_deferred_param_grads(True)
opt = torch.optim.Adam([{'params': ds.learnable_weights()}, {'params': [p for p in net.parameters() if p.requires_grad]}], lr=1e-4)
# monkeypatch to record shapes
recs=[]
orig_nt=me._gemm_nt; orig_tn=me._gemm_tn
def nt(A,lda,B,ldb,C,ldc,M,N,K,bias,group,act,mode,**kw):
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True); e0.record(); orig_nt(A,lda,B,ldb,C,ldc,M,N,K,bias,group,act,mode,**kw); e1.record()
    recs.append(('NT',M,N,K,group,mode,e0,e1))
def tn(Z,ldz,A,lda,R,N,K,lddw,group=1,**kw):
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True); e0.record(); r=orig_tn(Z,ldz,A,lda,R,N,K,lddw,group,**kw); e1.record()
    recs.append(('TN',R,N,K,group,0,e0,e1)); return r
me._gemm_nt=nt; me._gemm_tn=tn
def step(it):
    fids = torch.tensor([(5 + 3*it) % 60, (17 + 3*it) % 60, (30 + 3*it) % 60], device='cuda:0')
    datas = ds.batch(fids); ratio = {'sdfRatio': 1., 'deformerRatio': it / 2500. + 0.5, 'renderRatio': 1.}
    opt.zero_grad(set_to_none=True)
    loss = net(datas, 2048, ratio, fids); loss.backward(); net.propagateTmpPsGrad(fids, ratio); opt.step()
for it in range(3): step(it)
recs.clear()
for it in range(3, 9): step(it)
torch.cuda.synchronize()
agg=collections.defaultdict(lambda:[0,0.0,0.0])
for k,M,N,K,g,mode,e0,e1 in recs:
    mb = 'M<8k' if M<8192 else ('M<32k' if M<32768 else ('M<128k' if M<131072 else 'M>=128k'))
    key=(k,mb,N,K,g,mode); a=agg[key]; a[0]+=1; a[1]+=e0.elapsed_time(e1); a[2]+=2.0*M*N*K
tot=sum(a[1] for a in agg.values())
print('total gemm ms/iter', tot/6)
for key,a in sorted(agg.items(), key=lambda kv:-(kv[1][1]-kv[1][2]/125e9))[:30]:
    lost = a[1]/6 - a[2]/6/125e9          # ms per iteration above what 125 TFLOP/s would take
    print(f"{str(key):55s} calls/it {a[0]/6:6.1f}  ms/it {a[1]/6:7.2f}  TF/s {a[2]/a[1]/1e9:6.1f}  over-125TF {lost:6.2f} ms")
