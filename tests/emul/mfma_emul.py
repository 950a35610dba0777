"""CPU emulation of the wave64 MFMA dataflow used by csrc/field.hip (development/test aid, not product code).

Semantics from /opt/skills/guides/cdna_hip_programming.md section 3:
  v_mfma_f32_32x32x16_f16 : lane l holds A[i=l&31][k=kmap(l>>5,e)], B[k=kmap(l>>5,e)][j=l&31], e<8
  v_mfma_f32_32x32x2_f32  : lane l holds A[i=l&31][k=l>>5],        B[k=l>>5][j=l&31]
  C/D (both)              : lane l, reg r holds D[row=(r&3)+8*(r>>2)+4*(l>>5)][col=l&31]
The k-slot -> k mapping inside one instruction is irrelevant as long as A and B use the same one; the emulator
uses a scrambled kmap on purpose so that any reliance on a particular mapping shows up as a wrong result.
"""
import numpy as np

KMAP16 = np.array([[3, 9, 0, 14, 6, 11, 5, 12], [1, 15, 8, 2, 13, 4, 10, 7]])  # [h][e] -> k (a permutation of 0..15)


def row_of(r, h):
    return (r & 3) + 8 * (r >> 2) + 4 * h


def mfma_32x32x16(a, b, c):
    """a,b: [64 lanes][8]; c: [64][16] -> d [64][16]"""
    A = np.zeros((32, 16)); B = np.zeros((16, 32))
    for l in range(64):
        h = l >> 5
        for e in range(8):
            A[l & 31, KMAP16[h, e]] = a[l, e]
            B[KMAP16[h, e], l & 31] = b[l, e]
    D = A @ B
    d = c.copy()
    for l in range(64):
        for r in range(16):
            d[l, r] += D[row_of(r, l >> 5), l & 31]
    return d


def mfma_32x32x2(a, b, c):
    """a,b: [64]; c: [64][16]"""
    A = np.zeros((32, 2)); B = np.zeros((2, 32))
    for l in range(64):
        A[l & 31, l >> 5] = a[l]; B[l >> 5, l & 31] = b[l]
    D = A @ B
    d = c.copy()
    for l in range(64):
        for r in range(16):
            d[l, r] += D[row_of(r, l >> 5), l & 31]
    return d


# ---- index formulas shared (by construction) with csrc/field.hip ------------------------------------------
def kphys_f16(layer, s, h, e):
    """physical input index feeding logical K-slot (s,h,e) of an f16 layer"""
    if layer == 0:
        return 16 * h + 8 * s + e                      # kind X of field.hip: lane-half h holds features 16h .. 16h+15
    return 32 * (s >> 1) + row_of(8 * (s & 1) + e, h)  # kind D: neuron held in acc[s>>1][8*(s&1)+e] of this lane


def kphys_f32(layer, s, h):
    if layer == 0:
        return 16 * h + s if s < 16 else 32 + 16 * h + (s - 16)   # kind X, one value per step
    return 32 * (s >> 4) + row_of(s & 15, h)


def mlp_f16(feat, Ws, Bs):
    """feat [32 points][K0]; Ws[l] [out][in]; returns [32 points][out_last] computed the way the kernel does."""
    nl = len(Ws)
    acts = None
    for layer, (W, bia) in enumerate(zip(Ws, Bs)):
        out_dim, in_dim = W.shape
        kin = feat.shape[1] if layer == 0 else in_dim
        kin_pad = (kin + 15) // 16 * 16
        nt = (out_dim + 31) // 32
        acc = []
        for t in range(nt):
            c = np.zeros((64, 16))
            for l in range(64):
                for r in range(16):
                    o = 32 * t + row_of(r, l >> 5)
                    c[l, r] = bia[o] if o < out_dim else 0.0
            for s in range(kin_pad // 16):
                a = np.zeros((64, 8)); b = np.zeros((64, 8))
                for l in range(64):
                    i, h = l & 31, l >> 5
                    for e in range(8):
                        k = kphys_f16(layer, s, h, e)
                        o = 32 * t + i
                        a[l, e] = W[o, k] if (o < out_dim and k < in_dim) else 0.0
                        if layer == 0:
                            b[l, e] = feat[l & 31, k] if k < kin else 0.0
                        else:
                            b[l, e] = max(acts[s >> 1][l, 8 * (s & 1) + e], 0.0)  # relu of previous acc regs
                c = mfma_32x32x16(a, b, c)
            acc.append(c)
        acts = acc
    out = np.zeros((32, Ws[-1].shape[0]))
    for l in range(32):  # lanes with h == 0 hold rows 0..3
        for r in range(Ws[-1].shape[0]):
            out[l, r] = acts[0][l, r]
    return out


def mlp_f32(feat, Ws, Bs):
    acts = None
    for layer, (W, bia) in enumerate(zip(Ws, Bs)):
        out_dim, in_dim = W.shape
        nt = (out_dim + 31) // 32
        acc = []
        for t in range(nt):
            c = np.zeros((64, 16))
            for l in range(64):
                for r in range(16):
                    o = 32 * t + row_of(r, l >> 5)
                    c[l, r] = bia[o] if o < out_dim else 0.0
            for s in range(in_dim // 2):
                a = np.zeros(64); b = np.zeros(64)
                for l in range(64):
                    i, h = l & 31, l >> 5
                    k = kphys_f32(layer, s, h)
                    o = 32 * t + i
                    a[l] = W[o, k] if o < out_dim else 0.0
                    b[l] = feat[l & 31, k] if layer == 0 else max(acts[s >> 4][l, s & 15], 0.0)
                c = mfma_32x32x2(a, b, c)
            acc.append(c)
        acts = acc
    out = np.zeros((32, Ws[-1].shape[0]))
    for l in range(32):
        for r in range(Ws[-1].shape[0]):
            out[l, r] = acts[0][l, r]
    return out


def ref_mlp(feat, Ws, Bs):
    h = feat
    for l, (W, b) in enumerate(zip(Ws, Bs)):
        h = h @ W.T + b
        if l != len(Ws) - 1:
            h = np.maximum(h, 0)
    return h


def check(seed=0):
    rng = np.random.default_rng(seed)
    for (F, H, NL) in [(32, 64, 3), (32, 32, 2)]:
        dims = [F] + [H] * (NL - 1) + [4]
        Ws = [rng.normal(size=(dims[i + 1], dims[i])) for i in range(NL)]
        Bs = [rng.normal(size=dims[i + 1]) for i in range(NL)]
        feat = rng.normal(size=(32, F))
        ref = ref_mlp(feat, Ws, Bs)
        e16 = np.abs(mlp_f16(feat, Ws, Bs) - ref).max()
        e32 = np.abs(mlp_f32(feat, Ws, Bs) - ref).max()
        assert e16 < 1e-9 and e32 < 1e-9, (F, H, NL, e16, e32)
    return True


if __name__ == "__main__":
    check()
    print("MFMA chaining index math OK")
