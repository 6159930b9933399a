import numpy as np, torch, sys
sys.path.insert(0,'.')
from tests import util
from delora_amd import geometry as G
g = util.load_golden("normals_mid")
image = torch.from_numpy(g["image"]).cuda()
got = G.normals(image, 3, 5, 0.5, 10)[0].cpu().numpy()
img = g["image"][0,:3].astype(np.float64); H,W = img.shape[1:]
v,u = g["v"], g["u"]; ref=g["normals"].astype(np.float64); has=g["has"]
# fp64 truth with fp32 gate semantics
img32 = g["image"][0,:3]
r32 = torch.norm(torch.from_numpy(img32).view(1,3,-1),dim=1)[0].numpy().reshape(H,W)
truth = np.zeros_like(ref); cnt=np.zeros(len(v),int)
for k in range(len(v)):
    vs = np.clip(np.arange(v[k]-3, v[k]+4),0,H-1); us=np.clip(np.arange(u[k]-5,u[k]+6),0,W-1)
    VV,UU = np.meshgrid(vs,us,indexing="ij")
    nb = img[:,VV,UU].reshape(3,-1); rr = r32[VV,UU].reshape(-1)
    ok = ~(np.abs(rr - r32[v[k],u[k]]) > np.float32(0.5)) & (nb!=0).any(0)
    cnt[k]=ok.sum()
    if ok.sum()>=10:
        P = nb[:,ok]; C = np.cov(P)
        w,V = np.linalg.eigh(C); n=V[:,0]
        if n@img[:,v[k],u[k]]>0: n=-n
        truth[k]=n
gotl = got[:,v,u].T.astype(np.float64)
both = has & (np.abs(gotl).sum(1)>0) & (np.abs(truth).sum(1)>0)
def ang(a,b): return np.arccos(np.clip((a*b).sum(1),-1,1))
ag = ang(gotl[both],truth[both]); ar = ang(ref[both],truth[both])
print("count mismatch vs fixture count:", (cnt!=g["count"]).sum())
print("GPU vs fp64 truth: median %.2e p99 %.2e max %.2e"%(np.median(ag),np.percentile(ag,99),ag.max()))
print("REF vs fp64 truth: median %.2e p99 %.2e max %.2e"%(np.median(ar),np.percentile(ar,99),ar.max()))
lam=g["eigenvalues"][both].astype(np.float64); gap=(lam[:,1]-lam[:,0])/lam[:,2]
for lo,hi in ((1e-3,1e-2),(1e-2,1e-1),(1e-1,1)):
    m=(gap>lo)&(gap<=hi)
    if m.sum(): print("gap (%g,%g] n=%d  ref med %.2e max %.2e | gpu med %.2e max %.2e"%(lo,hi,m.sum(),np.median(ar[m]),ar[m].max(),np.median(ag[m]),ag[m].max()))
