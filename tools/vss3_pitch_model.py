"""Which pitch for the compact vectorscope table (HASH 3, scope_device.h)?  A model of the table's LDS bank traffic: sampled wave
instructions of scope_trio_kernel (4 image rows H / 4 apart x 16 column pairs, texel 0 of every lane) on synthetic noise / natural /
ramp / concentrated frames pushed through the oracle's K0; per instruction the worst bank's cost (one unit per distinct address, 0.7
more per further lane on an address: profiles/round1/lds_atomic_microbench.txt), averaged.  CPU only; output kept as
profiles/round5/vss3_pitch_model.txt.  Test infrastructure (it calls the oracle); never part of the product."""
import sys, numpy as np
import os; R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'oracle'))
import scope_oracle
o = scope_oracle.COracle()
W,H=3840,2160
rng=np.random.default_rng(1)
def ramp():
    xs=np.arange(W)[None,:]; ys=np.arange(H)[:,None]
    n=rng.integers(0,4,(H,W,3))
    f=np.empty((H,W,4),np.uint8)
    f[...,0]=np.clip(xs*255//W+n[...,0],0,255); f[...,1]=np.clip(ys*255//H+n[...,1],0,255)
    f[...,2]=np.clip((xs+ys)*255//(W+H)+n[...,2],0,255); f[...,3]=255
    return f.reshape(H,W*4)
def natural():
    # smooth field: bilinear upsample of 36x64 control points + noise
    ctrl=rng.random((37,65,3))*255
    yy=np.linspace(0,36,H,endpoint=False); xx=np.linspace(0,64,W,endpoint=False)
    y0=yy.astype(int); x0=xx.astype(int); fy=(yy-y0)[:,None,None]; fx=(xx-x0)[None,:,None]
    img=(ctrl[y0][:,x0]*(1-fy)*(1-fx)+ctrl[y0+1][:,x0]*fy*(1-fx)+ctrl[y0][:,x0+1]*(1-fy)*fx+ctrl[y0+1][:,x0+1]*fy*fx)
    img=img+rng.integers(0,4,(H,W,3))-1.5
    f=np.empty((H,W,4),np.uint8); f[...,:3]=np.clip(np.round(img),0,255); f[...,3]=255
    return f.reshape(H,W*4)
def noise():
    f=rng.integers(0,256,(H,W,4),dtype=np.uint8); f[...,3]=255; return f.reshape(H,W*4)
def layouts(u,v):
    out={}
    w3=(v-16)*116+((u-8)>>1); out['hash3 pitch116']=w3
    out['hash3 + xor (v>>3)&3']=w3 ^ ((v>>3)&3)
    out['pitch 120']=(v-16)*120+((u-8)>>1)
    out['pitch 117(odd)']=(v-16)*117+((u-8)>>1)
    out['pitch 124']=(v-16)*124+((u-8)>>1)
    out['pitch116 xor v&3']=w3 ^ (v&3)
    out['hash0 xor8(v&15)']=v*128+((u>>1)^(8*(v&15)))
    return out
def cost(words):
    # words: [ninstr, 64]; returns mean over instr of max lanes per bank, and of max same-address multiplicity
    banks=words&31
    mb=np.zeros(len(words)); ma=np.zeros(len(words))
    for i in range(len(words)):
        mb[i]=np.bincount(banks[i],minlength=32).max()
        ma[i]=np.unique(words[i],return_counts=True)[1].max()
    return mb.mean(), ma.mean()
for name,gen in ():
    rgb=gen(); yuv=o.rgb_to_yuv(rgb,W,2).reshape(H,W,4)
    U=yuv[...,0].astype(np.int64); V=yuv[...,2].astype(np.int64)
    # sample wave instructions: strip s, wave wv, slot g: rows rsub*540+wv+16g ; lanes (rsub, pair) ; texel p0 (even col)
    samples=[]
    for _ in range(400):
        s=rng.integers(0,120); wv=rng.integers(0,16); g=rng.integers(0,33)
        rows=np.array([r*540+wv+16*g for r in range(4)]); cols=s*32+2*np.arange(16)
        samples.append((rows,cols))
    print(name)
    for lay in layouts(np.int64(100),np.int64(100)).keys():
        ws=[]
        for rows,cols in samples:
            u=U[rows][:,cols].reshape(-1); v=V[rows][:,cols].reshape(-1)
            ws.append(layouts(u,v)[lay])
        mb,ma=cost(np.array(ws))
        print(f"   {lay:24s} max lanes per bank {mb:5.2f}   max same-address {ma:5.2f}")

print("---- pitch scan")
def conc():
    f=rng.integers(120,136,(H,W,4),dtype=np.uint8); f[...,3]=255; return f.reshape(H,W*4)
conts={}
for name,gen in (('noise',noise),('natural',natural),('ramp',ramp),('conc',conc)):
    rgb=gen(); yuv=o.rgb_to_yuv(rgb,W,2).reshape(H,W,4)
    conts[name]=(yuv[...,0].astype(np.int64),yuv[...,2].astype(np.int64))
samples=[]
for _ in range(300):
    s=rng.integers(0,120); wv=rng.integers(0,16); g=rng.integers(0,33)
    samples.append((np.array([r*540+wv+16*g for r in range(4)]), s*32+2*np.arange(16)))
def score(fn):
    out=[]
    for name,(U,V) in conts.items():
        tot=0
        for rows,cols in samples:
            u=U[rows][:,cols].reshape(-1); v=V[rows][:,cols].reshape(-1)
            w=fn(u,v)
            # cost model: per bank, sum over addresses: distinct address 1 unit each, same-address extra lanes 1.7 units
            b=w&31
            best=0
            for bank in np.unique(b):
                ws=w[b==bank]
                _,cnt=np.unique(ws,return_counts=True)
                c=len(cnt)+0.7*(cnt-1).sum()
                best=max(best,c)
            tot+=best
        out.append(tot/len(samples))
    return out
print("names", list(conts))
for pitch in (116,117,118,119,120,121,122,123,124,125,126,127):
    print(pitch, ["%.2f"%x for x in score(lambda u,v:(v-16)*pitch+((u-8)>>1))])
print("116 xor v>>3&3", ["%.2f"%x for x in score(lambda u,v:((v-16)*116+((u-8)>>1))^((v>>3)&3))])
print("116 xor v>>2&3", ["%.2f"%x for x in score(lambda u,v:((v-16)*116+((u-8)>>1))^((v>>2)&3))])
