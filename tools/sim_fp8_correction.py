"""Developer tool (NumPy only): would fp8 (e4m3) operands do for the two hi x lo correction products of the fp16 hi/lo
split used by kernels 2c/2d/2e?  Emulates the three schemes on BASELINE's 16x16 array, 8 Wheel foci, 40 000 random
voxels and prints the normalised max field error of each (gate: 1e-5).  Result recorded in DESIGN.md section 9."""
import numpy as np
rng=np.random.default_rng(3)
def q_e4m3(x):
    x=np.asarray(x,dtype=np.float64); s=np.sign(x); a=np.abs(x)
    a=np.minimum(a,448.0)
    e=np.floor(np.log2(np.maximum(a,1e-300)))
    e=np.maximum(e,-6)            # subnormals share the exponent of the smallest normal
    step=2.0**(e-3)
    return s*np.round(a/step)*step
def f16(x): return np.asarray(x,dtype=np.float64).astype(np.float16).astype(np.float64)
def split16(z):
    hr,hi=f16(z.real),f16(z.imag); lr,li=f16(z.real-hr),f16(z.imag-hi)
    return hr+1j*hi, lr+1j*li
def q8c(z,scale): return (q_e4m3(z.real*scale)+1j*q_e4m3(z.imag*scale))/scale
# geometry: 16x16 array pitch 3 mm, f0 400 kHz, c 1500; voxels random in 64 mm cube z from 5 mm; 8 foci on wheel
N=16; pitch=3e-3; xe=(np.arange(N)-(N-1)/2)*pitch
ex,ey=np.meshgrid(xe,xe,indexing='ij'); epos=np.stack([ex.ravel(),ey.ravel(),np.zeros(N*N)],1)
f0=400e3;c=1500.;lam=c/f0;k=2*np.pi/lam
V=40000
vox=np.stack([rng.uniform(-32e-3,32e-3,V),rng.uniform(-32e-3,32e-3,V),rng.uniform(5e-3,69e-3,V)],1)
foci=np.array([[5e-3*np.cos(t),5e-3*np.sin(t),40e-3] for t in 2*np.pi*np.arange(7)/63]+[[0,0,40e-3]])
vox=np.vstack([vox,foci])
d=np.linalg.norm(vox[:,None,:]-epos[None,:,:],axis=2)           # [V,E]
G=(lam/d)*np.exp(1j*k*d)*0.05                                       # scaled like g_scale so |G| ~ 0.05..1 
df=np.linalg.norm(foci[:,None,:]-epos[None,:,:],axis=2); tof=df/c; tau=tof.max(1,keepdims=True)-tof
W=np.exp(1j*2*np.pi*f0*tau).T                                     # [E,F], |W| = 1
P=G@W; ref=np.abs(P).max()
def err(Pa): return np.abs(Pa-P).max()/ref
Gh,Gl=split16(G); Wh,Wl=split16(W)
print("hi only            ",err(Gh@Wh))
print("3 products (now)   ",err(Gh@Wh+Gl@Wh+Gh@Wl))
for name,(sGl,sWh,sGh,sWl) in {"fp8 corr, scales 2^17/2^6/2^6/2^17":(2.0**17,2.0**6,2.0**6,2.0**17),
                               "fp8 corr, scales 2^19/2^8/2^8/2^19":(2.0**19,2.0**8,2.0**8,2.0**19)}.items():
    c1=q8c(Gl,sGl)@q8c(Wh,sWh); c2=q8c(Gh,sGh)@q8c(Wl,sWl)
    print(name, err(Gh@Wh+c1+c2), " max|Gl|*s",np.abs(Gl.real).max()*sGl, "max|Wl|*s",np.abs(Wl.real).max()*sWl)
# fp16 hi x fp16 hi + ONE fp16 correction (Gl*Wh only) for reference
print("2 products Gl*Wh   ",err(Gh@Wh+Gl@Wh))
print("2 products Gh*Wl   ",err(Gh@Wh+Gh@Wl))
