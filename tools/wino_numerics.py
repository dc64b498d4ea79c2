import numpy as np
rng = np.random.default_rng(0)
def split(x):
    x = x.astype(np.float32)
    hi = x.astype(np.float16)
    lo = (x - hi.astype(np.float32)).astype(np.float16)
    return hi.astype(np.float64), lo.astype(np.float64)
def x3dot(A, B):
    # A [M,K], B [N,K] fp32 -> emulated x3 product with fp32 accumulation in K chunks of 16
    ah, al = split(A); bh, bl = split(B)
    M, K = A.shape; N = B.shape[0]
    acc = np.zeros((M, N), np.float32)
    for k in range(0, K, 16):
        s = slice(k, k + 16)
        for (a, b) in ((al, bh), (ah, bl), (ah, bh)):
            acc = (acc.astype(np.float64) + a[:, s] @ b[:, s].T).astype(np.float32)
    return acc
# conv 1-D along W with Cin channels, 3 taps (ky folded into channel dim: K = 3*Cin per tap)
Cin, Cout, W, Bn = 512*3, 64, 68, 6     # channels x ky folded
x = np.maximum(rng.standard_normal((Bn, W + 2, Cin)), 0).astype(np.float32)   # post-ReLU, padded by 1 each side
x[:, 0] = 0; x[:, -1] = 0
g = (rng.standard_normal((Cout, 3, Cin)) * np.sqrt(2.0 / (9 * 512))).astype(np.float32)
wscale = 2.0 ** np.floor(np.log2(16384 / np.abs(g).max()))
ref = np.zeros((Bn, W, Cout))
for t in range(3):
    ref += x[:, t:t + W].astype(np.float64) @ g[:, t].astype(np.float64).T
def direct():
    out = np.zeros((Bn * W, Cout), np.float32)
    A = np.concatenate([x[:, t:t + W].reshape(Bn * W, Cin) for t in range(3)], 1)
    Bm = np.concatenate([g[:, t] for t in range(3)], 1) * wscale
    return (x3dot(A, Bm.astype(np.float32)) / wscale).reshape(Bn, W, Cout)
def wino(BT, G, AT, r):
    n = BT.shape[0]
    T = W // r
    # tiles: inputs x[r*t : r*t+n]
    d = np.stack([x[:, r * t:r * t + n] for t in range(T)], 1)       # [Bn,T,n,Cin]
    V = np.einsum('ij,btjc->btic', BT.astype(np.float32), d).astype(np.float32)   # fp32 transform (einsum in fp32)
    U = np.einsum('ij,ojc->oic', G, g.astype(np.float64))            # float64 weight transform
    Y = np.zeros((Bn, T, r, Cout), np.float64)
    for m in range(n):
        Um = U[:, m]
        s = 2.0 ** np.floor(np.log2(16384 / np.abs(Um).max()))
        Mm = x3dot(V[:, :, m].reshape(Bn * T, Cin), (Um * s).astype(np.float32)) / np.float32(s)
        Mm = Mm.reshape(Bn, T, Cout)
        for rr in range(r):
            Y[:, :, rr] = (Y[:, :, rr].astype(np.float32) + np.float32(AT[rr, m]) * Mm).astype(np.float32)
    return Y.reshape(Bn, W, Cout)
BT2 = np.array([[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]], float)
G2 = np.array([[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]])
AT2 = np.array([[1,1,1,0],[0,1,-1,-1]], float)
BT4 = np.array([[4,0,-5,0,1,0],[0,-4,-4,1,1,0],[0,4,-4,-1,1,0],[0,-2,-1,2,1,0],[0,2,-1,-2,1,0],[0,4,0,-5,0,1]], float)
G4 = np.array([[1/4,0,0],[-1/6,-1/6,-1/6],[-1/6,1/6,-1/6],[1/24,1/12,1/6],[1/24,-1/12,1/6],[0,0,1]])
AT4 = np.array([[1,1,1,1,1,0],[0,1,-1,2,-2,0],[0,1,1,4,4,0],[0,1,-1,8,-8,1]], float)
sc = np.abs(ref).max(); rms = np.sqrt((ref**2).mean())
for name, y in (("direct x3", direct()), ("F(2,3)", wino(BT2, G2, AT2, 2)), ("F(4,3)", wino(BT4, G4, AT4, 4))):
    e = y - ref
    print(f"{name:10s} max|err|/max|ref| = {np.abs(e).max()/sc:.3e}   rms err / rms ref = {np.sqrt((e**2).mean())/rms:.3e}")
# plain fp32 direct (torch-like) for scale
A = np.concatenate([x[:, t:t + W].reshape(Bn * W, Cin) for t in range(3)], 1); Bm = np.concatenate([g[:, t] for t in range(3)], 1)
y32 = (A @ Bm.T).reshape(Bn, W, Cout)
e = y32 - ref
print(f"fp32 numpy  max|err|/max|ref| = {np.abs(e).max()/sc:.3e}   rms = {np.sqrt((e**2).mean())/rms:.3e}")
