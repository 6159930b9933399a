#!/usr/bin/env python3
"""Index-level emulation (numpy, no GPU) of tools/exp/wino_wgrad.hip: thread roles, LDS layout with its 16-byte swizzle,
MFMA operand / accumulator lane mapping (the one csrc/wino.hip relies on), partial write and the reduce -- against the direct
weight gradient.  Validates the addressing of the draft kernel; rounding and performance need the GPU."""
import numpy as np

PLANE = 528
rng = np.random.default_rng(1)


def run(N, H, W, C, K, nslabs_want=3):
    x = rng.normal(size=(N, H, W, C)).astype(np.float64)
    g = rng.normal(size=(N, H, W, K)).astype(np.float64)
    th, tw8 = H // 2, (W // 2) // 8
    total = N * th * tw8
    nslabs = min(nslabs_want, total)
    per = -(-total // nslabs)
    nslabs = -(-total // per)
    ws = np.zeros((nslabs, 16, K, C))
    for kt in range(K // 64):
        for ct in range(C // 64):
            for slab in range(nslabs):
                k0, c0 = kt * 64, ct * 64
                acc = np.zeros((8, 16, 64, 16))                  # [wave][plane-local... (xl)][lane][reg]  -> use dict-like arrays
                acc = np.zeros((8, 8, 64, 16))
                for ch in range(slab * per, min((slab + 1) * per, total)):
                    lds = np.zeros(2 * 16 * PLANE)
                    u = ch
                    b8 = u % tw8; u //= tw8
                    ta = u % th
                    n = u // th
                    # raw patch by "DMA": wave w brings pieces w, w+8, w+16 (clamped) of 4 pixels x 64 channels, 16 bytes per lane
                    raw = np.full(72 * 64, np.nan)
                    for wave in range(8):
                        for it in range(3):
                            piece = min(wave + 8 * it, 17)
                            for lane in range(64):
                                pix = piece * 4 + (lane >> 4)
                                pi, pj = divmod(pix, 18)
                                row = min(max(2 * ta - 1 + pi, 0), H - 1)
                                col = (16 * b8 - 1 + pj) % W
                                cq = lane & 15
                                raw[piece * 256 + lane * 4:piece * 256 + lane * 4 + 4] = x[n, row, col, c0 + cq * 4:c0 + cq * 4 + 4]
                    assert not np.isnan(raw).any()
                    for tid in range(512):
                        r, tq, bcol = tid & 63, (tid >> 6) & 1, tid >> 7
                        j0, j1 = (0 if bcol == 0 else 1), (3 if bcol == 3 else 2)
                        sg0, sg1 = (-1.0 if bcol == 2 else 1.0), (-1.0 if bcol in (0, 3) else 1.0)
                        w_off = r * 8 + ((tq ^ ((r >> 3) & 1)) * 4)
                        vD, vG = np.zeros((4, 4)), np.zeros((4, 4))
                        for e in range(4):
                            tb = b8 * 8 + tq * 4 + e
                            tt = []
                            for i in range(4):
                                row = 2 * ta - 1 + i
                                mask = 1.0 if 0 <= row < H else 0.0
                                d0 = raw[(i * 18 + 2 * (4 * tq + e) + j0) * 64 + r] * mask
                                d1 = raw[(i * 18 + 2 * (4 * tq + e) + j1) * 64 + r] * mask
                                tt.append(sg0 * d0 + sg1 * d1)
                            vD[0, e], vD[1, e], vD[2, e], vD[3, e] = tt[0] - tt[2], tt[1] + tt[2], tt[2] - tt[1], tt[1] - tt[3]
                            h = []
                            for p in range(2):
                                g0, g1 = g[n, 2 * ta + p, 2 * tb, k0 + r], g[n, 2 * ta + p, 2 * tb + 1, k0 + r]
                                h.append([g0, g0 + g1, g0 - g1, -g1][bcol])
                            vG[0, e], vG[1, e], vG[2, e], vG[3, e] = h[0], h[0] + h[1], h[0] - h[1], -h[1]
                        for i in range(4):
                            lds[(i * 4 + bcol) * PLANE + w_off:(i * 4 + bcol) * PLANE + w_off + 4] = vG[i]
                            o = 16 * PLANE + (i * 4 + bcol) * PLANE + w_off
                            lds[o:o + 4] = vD[i]
                    for wave in range(8):
                        mb, nb, xh = wave & 1, (wave >> 1) & 1, wave >> 2
                        A = np.zeros((8, 64, 4)); Bv = np.zeros((8, 64, 4))
                        for lane in range(64):
                            li, half = lane & 31, lane >> 5
                            arow, brow = mb * 32 + li, nb * 32 + li
                            a_off = arow * 8 + ((half ^ ((arow >> 3) & 1)) * 4)
                            b_off = 16 * PLANE + brow * 8 + ((half ^ ((brow >> 3) & 1)) * 4)
                            for xl in range(8):
                                A[xl, lane] = lds[(xh * 8 + xl) * PLANE + a_off:(xh * 8 + xl) * PLANE + a_off + 4]
                                Bv[xl, lane] = lds[(xh * 8 + xl) * PLANE + b_off:(xh * 8 + xl) * PLANE + b_off + 4]
                        for xl in range(8):
                            for j in range(4):
                                # v_mfma_f32_32x32x2: lane l supplies A[m = l%32][kk = l/32] and B[kk = l/32][n = l%32]
                                Am = np.zeros((32, 2)); Bm = np.zeros((2, 32))
                                for lane in range(64):
                                    Am[lane & 31, lane >> 5] = A[xl, lane, j]
                                    Bm[lane >> 5, lane & 31] = Bv[xl, lane, j]
                                D = Am @ Bm
                                for lane in range(64):
                                    for q in range(16):
                                        acc[wave, xl, lane, q] += D[8 * (q // 4) + 4 * (lane >> 5) + (q % 4), lane & 31]
                for wave in range(8):
                    mb, nb, xh = wave & 1, (wave >> 1) & 1, wave >> 2
                    for xl in range(8):
                        for lane in range(64):
                            li, half = lane & 31, lane >> 5
                            for q in range(16):
                                m = mb * 32 + 8 * (q // 4) + 4 * half + (q % 4)
                                ws[slab, xh * 8 + xl, k0 + m, c0 + nb * 32 + li] = acc[wave, xl, lane, q]
    U = ws.sum(axis=0).reshape(4, 4, K, C)
    Gt = np.array([[1, 0.5, 0.5, 0], [0, 0.5, -0.5, 0], [0, 0.5, 0.5, 1]])
    dw = np.einsum("ra,abkc,sb->krsc", Gt, U, Gt)
    xp = np.zeros((N, H + 2, W + 2, C)); xp[:, 1:-1, 1:-1] = x; xp[:, 1:-1, 0] = x[:, :, -1]; xp[:, 1:-1, -1] = x[:, :, 0]
    ref = np.zeros((K, 3, 3, C))
    for rr in range(3):
        for ss in range(3):
            ref[:, rr, ss, :] = np.einsum("nijk,nijc->kc", g, xp[:, rr:rr + H, ss:ss + W])
    err = np.abs(dw - ref).max() / np.abs(ref).max()
    print(f"N={N} H={H} W={W} C={C} K={K}: slabs {nslabs}, relative error {err:.2e}", "ok" if err < 1e-12 else "MISMATCH")
    return err < 1e-12


ok = run(1, 4, 32, 64, 64) and run(1, 2, 16, 64, 128, nslabs_want=1)
raise SystemExit(0 if ok else 1)
