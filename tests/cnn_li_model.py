"""Numpy statement of the lane = image formulation of the CNN front end (conv 3x3 -> conv 3x3 -> pool -> conv 3x3 -> pool per
channel, BitNetMCU_MNIST_dll.c:48-91 / BitNetMCU_inference.c:238-322) as THREE Toeplitz products per channel on
v_mfma_i32_32x32x32_i8 with the images of a 32-image tile along the MFMA's N dimension - the data flow of
bitnetmcu_amd/csrc/bnm_cnn_li.hip, written down where it can be checked against the oracle without a GPU
(tests/test_cnn_li_model.py).  Everything the kernel and its host-side fragment builder must agree on lives here:

  D layout      lane (j, h) of v_mfma_i32_32x32x32_i8 holds, in D register i (0..15), row  rho(i, h) = (i & 3) + 8 (i >> 2) + 4 h
                of image (column) j.
  A / B layout  lane (r, kh) holds K slots 16 kh .. 16 kh + 15 of row r (A) / of image r (B).

  stage 1  conv1, 16x16 int8 image -> 14x14.  K-step s = image rows 2s, 2s+1 (slot = 16 (row & 1) + col: the B operand is 16
           contiguous image bytes per lane).  Output ROW PAIR r (conv1 rows 2r, 2r+1; r = 0..6) = one D tile from K-steps r, r+1;
           D register i of half h = position (row 2r + h, col i), i < 14.  The kernel being translation invariant, every row pair
           uses the same two A fragments.
           epilogue: v = relu(sum) >> 4 (<= 9,072: 14 bits) -> TWO int8 planes  lo = ((v & 255) ^ 128) as int8 = (v & 255) - 128,
           hi = v >> 8;  v = lo + 128 + 256 hi.  D register i -> byte i of the lane's 16-byte operand: half h of row pair r IS
           lane half h of conv2's K-step r (conv1 rows 2r, 2r+1; slot = 16 (row & 1) + col, cols 14, 15 padding with zero weights).
  stage 2  conv2, 14x14 -> 12x12 -> pool -> 6x6.  Output row pair r' (rows 2r', 2r'+1; r' = 0..5) from K-steps r', r'+1, per plane
           (two accumulators: the plane weight 256 does not fit an int8 weight).  D rows ordered so that register quad t of half h
           is pooling window q = 2t + h (t = 0..2; quad 3 padding): element e of the quad = position (2r' + (e >> 1), 2q + (e & 1)).
           epilogue: s_e = accL_e + 256 accH_e (+ 128 sum(w), the same for all four: added after the maximum);
           P = relu(max_e s_e + 128 sum(w)) >> 4  (<= 648,270: 20 bits) -> THREE planes  b0 = (P & 255) - 128, b1 = ((P >> 8) & 255) - 128,
           b2 = P >> 16;  P = b0 + 128 + 256 (b1 + 128) + 65536 b2.
           conv3's K: K-step 0 = pooled rows 0..3, K-step 1 = rows 4, 5;  slot = 16 h + 3 (row % 4) + t  <->  pooled (row, col 2t + h):
           a lane's own three windows of row pair r' are bytes 3 (r' % 4) + t of ITS half of the operand - no cross-lane traffic.
  stage 3  conv3, 6x6 -> 4x4 -> pool -> 2x2.  One D tile (16 of 32 rows) from K-steps 0, 1, per plane (three accumulators).
           Register quad t of half h (t = 0, 1) is pooling window u = 2t + h = (wr, wc) = (t, h); element e = position (2 wr + (e >> 1),
           2 wc + (e & 1)).  feature(c, u) = relu(max_e(acc0 + 256 acc1 + 65536 acc2) + (128 + 32768) sum(w)) >> 4.
  ReLUNorm over an image's 4 C features (BitNetMCU_inference.c:23-72) from a COMPRESSED per-(channel, image) record: the running
           maximum mx of the IMAGE's features so far (both lane halves: one k byte per image, 160 bytes of records per channel and
           wave); with sh = bitlength(mx >> 7) at the time of storing, k = max(sh - 1, 0) and f' = f >> k < 256 is exact for the
           final shift s >= sh:  (f + (1 << s >> 1)) >> s == (f' + (1 << (s - k) >> 1)) >> (s - k).
"""
import numpy as np


def d_row(i, h):
    return (i & 3) + 8 * (i >> 2) + 4 * h


# ---- Toeplitz fragments: dense [32 rows][2 K-steps x 32 slots] int8 per (stage, channel) ----------------------------------

def toeplitz_conv1(w9):
    """w9: the channel's 3x3 kernel, row-major (w[3 dy + dx]).  A[rho][32 s + slot], s = 0: image rows 2r, 2r+1; s = 1: 2r+2, 2r+3."""
    A = np.zeros((32, 64), np.int64)
    for h in range(2):
        for i in range(14):
            rho = d_row(i, h)
            for dy in range(3):
                for dx in range(3):
                    row = h + dy                      # relative to image row 2r
                    A[rho, 32 * (row >> 1) + 16 * (row & 1) + i + dx] += int(w9[3 * dy + dx])
    return A


def toeplitz_conv2(w9):
    """rows: quad t of half h = window q = 2t + h of row pair r'; K: conv1-output rows 2r' .. 2r'+3, slot = 16 (row & 1) + col."""
    A = np.zeros((32, 64), np.int64)
    for h in range(2):
        for t in range(3):
            q = 2 * t + h
            for e in range(4):
                rho = d_row(4 * t + e, h)
                orow, ocol = e >> 1, 2 * q + (e & 1)
                for dy in range(3):
                    for dx in range(3):
                        row = orow + dy
                        A[rho, 32 * (row >> 1) + 16 * (row & 1) + ocol + dx] += int(w9[3 * dy + dx])
    return A


def conv3_slot(prow, pcol):
    """K index (0..63) of pooled position (prow, pcol) of the 6x6 plane: K-step prow // 4, lane half pcol & 1, byte 3 (prow % 4) + pcol // 2."""
    return 32 * (prow >> 2) + 16 * (pcol & 1) + 3 * (prow & 3) + (pcol >> 1)


def toeplitz_conv3(w9):
    A = np.zeros((32, 64), np.int64)
    for h in range(2):
        for t in range(2):
            for e in range(4):
                rho = d_row(4 * t + e, h)
                orow, ocol = 2 * t + (e >> 1), 2 * h + (e & 1)
                for dy in range(3):
                    for dx in range(3):
                        A[rho, conv3_slot(orow + dy, ocol + dx)] += int(w9[3 * dy + dx])
    return A


def mfma(A32x32, B32xN):
    """D[rho][j] = sum_k A[rho][k] B[k][j]: one v_mfma_i32_32x32x32_i8 (exact in int32 for these magnitudes)."""
    assert np.abs(A32x32).max() <= 128 and B32xN.min() >= -128 and B32xN.max() <= 127
    return A32x32 @ B32xN


def lane_regs(D, h):
    """the 16 D registers of lane half h for every image: [16][N]"""
    return np.stack([D[d_row(i, h)] for i in range(16)])


def front_end_features(images, w1, w2, w3, pipelined=False):
    """images int8 [N][256]; w*: int8 [C][9].  Returns (features int64 [N][4C], records): the features as the reference orders
    them (channel-major, 2x2 pool output row-major) and the compressed per-(channel, lane half) records of the fused ReLUNorm.
    pipelined: stage 1's epilogue as cnn_li_fused_pipe_kernel computes it (bnm_cnn_li_tile_body_pipe.inc) - the accumulators start
    at -32768, v_cvt_pk_i16_i32's saturation is the ReLU; of u = relu(sum) - 32768 the high byte is the A plane (A - 128, A =
    relu(sum) >> 8) and (low byte & 0xF0) ^ 0x80 the B plane (16 B - 128, B = bits 4..7): conv1's output is 16 A + B, conv2's two
    sums combine as 256 dA + dB = 16 x the conv2 sum, the offsets travel in its bias.  Valid while no conv1 sum exceeds 65535 (asserted)."""
    N, C = images.shape[0], w1.shape[0]
    img = images.astype(np.int64).reshape(N, 16, 16)
    # B operands of stage 1: K-step s, slot 16 (row & 1) + col
    B1 = [np.concatenate([img[:, 2 * s].T, img[:, 2 * s + 1].T]) for s in range(8)]          # each [32][N]
    feats = np.zeros((N, 4 * C), np.int64)
    rec = np.zeros((C, 2, N, 3), np.int64)          # (f'_0, f'_1, k) per channel, lane half, image
    mx = np.zeros(N, np.int64)
    for c in range(C):
        T1, T2, T3 = toeplitz_conv1(w1[c]), toeplitz_conv2(w2[c]), toeplitz_conv3(w3[c])
        sw2, sw3 = int(w2[c].astype(np.int64).sum()), int(w3[c].astype(np.int64).sum())
        # ---- stage 1 -> conv2's operands, plane by plane ----
        lo = [np.zeros((32, N), np.int64) for _ in range(7)]
        hi = [np.zeros((32, N), np.int64) for _ in range(7)]
        for r in range(7):
            D = mfma(T1[:, :32], B1[r]) + mfma(T1[:, 32:], B1[r + 1])
            for h in range(2):
                regs = lane_regs(D, h)
                if pipelined:
                    assert regs[:14].max() <= 65535, "the pipelined form serves models whose conv1 sums stay below 2^16"
                    u = np.clip(regs[:14] - 32768, -32768, 32767)          # v_cvt_pk_i16_i32 (saturating): relu(sum) - 32768
                    # w = relu(sum) >> 4 = 16 A + B, A = relu(sum) >> 8, B = bits 4..7 of relu(sum)
                    lo[r][16 * h:16 * h + 14] = (u & 0xF0) - 128            # the B plane: (low byte & 0xF0) ^ 0x80 read as int8 = 16 B - 128
                    hi[r][16 * h:16 * h + 14] = u >> 8                      # the A plane: u's high byte as a signed byte = A - 128
                    hi[r][16 * h + 14:16 * h + 16] = -128                   # (padding halves hold 0x8000)
                else:
                    v = np.maximum(regs[:14], 0) >> 4
                    lo[r][16 * h:16 * h + 14] = (v & 255) - 128
                    hi[r][16 * h:16 * h + 14] = v >> 8
                lo[r][16 * h + 14:16 * h + 16] = -128          # (what the ^ 0x80 of a zero byte is; its weights are zero)
        # ---- stage 2 -> conv3's operands ----
        p = [[np.full((32, N), -128 if pl < 2 else 0, np.int64) for _ in range(2)] for pl in range(3)]
        for r2 in range(6):
            DL = mfma(T2[:, :32], lo[r2]) + mfma(T2[:, 32:], lo[r2 + 1])
            DH = mfma(T2[:, :32], hi[r2]) + mfma(T2[:, 32:], hi[r2 + 1])
            for h in range(2):
                rl, rh = lane_regs(DL, h), lane_regs(DH, h)
                for t in range(3):
                    s = rl[4 * t:4 * t + 4] + 256 * rh[4 * t:4 * t + 4]
                    if pipelined:      # s is 16 x the conv2 sum less the planes' offsets: (32768 + 128) x the weight sum
                        P = np.maximum(s.max(axis=0) + 257 * 128 * sw2, 0) >> 8
                    else:
                        P = np.maximum(s.max(axis=0) + 128 * sw2, 0) >> 4
                    k = conv3_slot(r2, 2 * t + h)
                    p[0][k >> 5][k & 31] = (P & 255) - 128
                    p[1][k >> 5][k & 31] = ((P >> 8) & 255) - 128
                    p[2][k >> 5][k & 31] = P >> 16
        # ---- stage 3 -> features ----
        acc = [mfma(T3[:, :32], p[pl][0]) + mfma(T3[:, 32:], p[pl][1]) for pl in range(3)]
        fh = []
        for h in range(2):
            r0, r1, r2_ = (lane_regs(a, h) for a in acc)
            f = []
            for t in range(2):
                s = r0[4 * t:4 * t + 4] + 256 * r1[4 * t:4 * t + 4] + 65536 * r2_[4 * t:4 * t + 4]
                f.append(np.maximum(s.max(axis=0) + (128 + 32768) * sw3, 0) >> 4)
                feats[:, 4 * c + 2 * t + h] = f[-1]
            fh.append(f)
        mx = np.maximum(mx, np.maximum(np.maximum(fh[0][0], fh[0][1]), np.maximum(fh[1][0], fh[1][1])))
        sh = np.array([int(m >> 7).bit_length() for m in mx])
        k = np.maximum(sh - 1, 0)
        for h in range(2):
            rec[c, h, :, 0], rec[c, h, :, 1], rec[c, h, :, 2] = fh[h][0] >> k, fh[h][1] >> k, k
    return feats, (rec, mx)


def relunorm_from_records(rec, mx):
    """The fused ReLUNorm from the compressed records: int8 [N][4C] act rows (BitNetMCU_inference.c:23-72 over all 4C features)."""
    C, _, N, _ = rec.shape
    s = np.array([int(v >> 7).bit_length() for v in mx])
    out = np.zeros((N, 4 * C), np.int64)
    for c in range(C):
        for h in range(2):
            k = rec[c, h, :, 2]
            assert (k <= np.maximum(s - 1, 0)).all() and (rec[c, h, :, :2] < 256).all()
            d = s - k
            rnd = (1 << d) >> 1
            for t in range(2):
                out[:, 4 * c + 2 * t + h] = np.minimum((rec[c, h, :, t] + rnd) >> d, 127)
    return out.astype(np.int8)
