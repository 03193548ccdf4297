"""Numpy model of the front-end kernel's FFT decomposition (index math check, no GPU needed).
2048-pt real FFT of a 512-sample frame = 1024-pt complex FFT of 256 packed samples, input-pruned
into 4 x 256-pt FFTs (k = 4m + r), each done as two in-register radix-16 passes."""
import numpy as np
rng=np.random.default_rng(0)
x=rng.standard_normal(512); w=0.5-0.5*np.cos(2*np.pi*np.arange(512)/512)
ref=np.fft.rfft(np.concatenate([x*w,np.zeros(1536)]))
W2048=np.exp(-2j*np.pi*np.arange(2048)/2048)
z=(x[0::2]*w[0::2])+1j*(x[1::2]*w[1::2])      # 256 complex
Z=np.zeros(1024,complex)
W16=np.exp(-2j*np.pi*np.arange(16)/16)
def fft16(v):
    t=np.zeros((4,4),complex)
    for j in range(4):
        x0,x1,x2,x3=v[j],v[j+4],v[j+8],v[j+12]
        t[j,0]=(x0+x2)+(x1+x3); t[j,1]=(x0-x2)-1j*(x1-x3); t[j,2]=(x0+x2)-(x1+x3); t[j,3]=(x0-x2)+1j*(x1-x3)
        for b in range(4): t[j,b]*=W16[(j*b)%16]
    out=np.zeros(16,complex)
    for b in range(4):
        x0,x1,x2,x3=t[0,b],t[1,b],t[2,b],t[3,b]
        out[b+0]=(x0+x2)+(x1+x3); out[b+4]=(x0-x2)-1j*(x1-x3); out[b+8]=(x0+x2)-(x1+x3); out[b+12]=(x0-x2)+1j*(x1-x3)
    return out
A=np.zeros((4,16,16),complex)
for r in range(4):
    for n0 in range(16):
        v=np.array([z[16*n1+n0]*W2048[(2*(16*n1+n0)*r)%2048] for n1 in range(16)])
        a=fft16(v)
        for m0 in range(16): A[r,n0,m0]=a[m0]*W2048[(8*n0*m0)%2048]
for r in range(4):
    for m0 in range(16):
        y=fft16(A[r,:,m0])
        for m1 in range(16): Z[4*(m0+16*m1)+r]=y[m1]
Zref=np.fft.fft(np.concatenate([z,np.zeros(768)]))
print('Z err',np.abs(Z-Zref).max())
k=np.arange(0,1025); Zk=Z[k%1024]; Zc=np.conj(Z[(1024-k)%1024])
X=0.5*(Zk+Zc)+W2048[k%2048]*(-0.5j)*(Zk-Zc)
print('X err',np.abs(X-ref).max(), np.abs(ref).max())
# torch.stft pads the window CENTRALLY to n_fft: the frame is 2048 samples with the Hann in
# [768,1280).  With center=True/reflect and hop 256 the 512 samples under the window are
# x[256t-256 .. 256t+255]; a time shift changes phase only, so |X|^2 is identical:
full=np.zeros(2048); full[768:1280]=x*w
print('power err vs centred',np.abs(np.abs(np.fft.rfft(full))**2-np.abs(ref)**2).max())
