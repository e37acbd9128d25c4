// Packed-weight layout of libcbgx (floats).  Built by cbgx_pack_weights from the reference
// state_dict tensors; consumed by every kernel.  One place defines it.
#pragma once
#include <cstddef>

namespace cbgx {

constexpr int H = 128;        // node_feat_dim
constexpr int HEADS = 16;     // n_heads
constexpr int DH = 8;         // H / HEADS
constexpr int G = 20;         // num_r_gaussian
constexpr int KNN = 32;       // k
constexpr int NT = 4;         // edge types
constexpr int GH = 160;       // gate MLP hidden (8 * G)
constexpr int KV_IN = 2 * H + NT + NT * G;  // 340
constexpr int PROW = 5 * H;   // node projection row: [PDk | PDv | PSk | PSv | q_hidden]

// ---- gate (denoiser.dist_emb.1) ----
constexpr size_t GATE_W1 = 0;                    // [160][20] row-major (out, in)
constexpr size_t GATE_B1 = GATE_W1 + GH * G;     // [160]
constexpr size_t GATE_LNG = GATE_B1 + GH;        // [160]
constexpr size_t GATE_LNB = GATE_LNG + GH;       // [160]
constexpr size_t GATE_W2 = GATE_LNB + GH;        // [160]
constexpr size_t GATE_B2 = GATE_W2 + GH;         // [1] (+3 pad)
// LDS image of the MFMA gate kernel: W1 fragments (centred) [10][5][64] | b1c | gamma | beta | w2 (160 each)
constexpr size_t GATE_IMG = GATE_B2 + 4;
constexpr size_t GATE_IMG_SIZE = (size_t)(GH / 16) * 5 * 64 + 4 * GH;   // 3840 floats
constexpr size_t GATE_SIZE = GATE_IMG + GATE_IMG_SIZE;

// ---- one attention block (x2h or h2x) ----
constexpr size_t A_WN = 0;                        // [128][640] K-major node projection
constexpr size_t A_BN = A_WN + (size_t)H * PROW;  // [640]
constexpr size_t A_WT = A_BN + PROW;              // [4][256]  type one-hot columns of W_a (k | v)
constexpr size_t A_WR = A_WT + NT * 2 * H;        // [4][20][256] rbf columns of W_a (k | v)
constexpr size_t A_LNK_G = A_WR + (size_t)NT * G * 2 * H;
constexpr size_t A_LNK_B = A_LNK_G + H;
constexpr size_t A_LNV_G = A_LNK_B + H;
constexpr size_t A_LNV_B = A_LNV_G + H;
constexpr size_t A_LNQ_G = A_LNV_B + H;
constexpr size_t A_LNQ_B = A_LNQ_G + H;
constexpr size_t A_WQ1T = A_LNQ_B + H;            // [128][128] K-major second q linear
constexpr size_t A_BQ1 = A_WQ1T + (size_t)H * H;  // [128]
constexpr size_t A_WBK = A_BQ1 + H;               // [128][128] (out n, in m) second k linear
constexpr size_t A_WBV = A_WBK + (size_t)H * H;   // x2h: [128 m][128 n] K-major; h2x: [16][128] (head, m)
constexpr size_t A_BBV = A_WBV + (size_t)H * H;   // [128] (h2x: first 16)
// LDS image of the MFMA edge kernel: one contiguous region copied verbatim into LDS.
//   frag_k [4 types][8 t][320]            split-f16 pieces of the rbf columns of W_a (k), A operand, edge-major.  A weight w
//                                         is carried as h = f16(w), l = f16(w - h); per (type, t) block, with h_s / l_s the
//                                         pieces of W[m(t, c)][g = 4s + q] of lane (c, q):
//                                           floats [  0,128): lane's (d0, d1) = [h0 h1 | h2 h3]
//                                           floats [128,256): lane's (d2, d3) = [h4 l0 | l1 l2]
//                                           floats [256,320): lane's  d4      = [l3 l4]
//                                         = 20 bytes per lane, the footprint of the five fp32 fragments it replaces
//   frag_v [4][8][320]                    same for v.  x2h: B operand, channel-major; h2x: A operand, edge-major
//   dwt    [2][256] (+512 unused)         dWt[lig_i][k|v] = Wt[type(src lig, i)] - Wt[type(src prot, i)]
//   ln     [4][128]                       gamma_k, beta_k, gamma_v, beta_v
//   wbv    x2h: [128][128]                second v Linear, rows (out n) of 16-byte chunks in the epilogue's order (edge_mfma.hip)
//          h2x: [8 t][64 lanes][4]        the 16 head rows as the B operand of the value contraction: lane (c = head, q) finds
//                                         Wbv[c][16 t + 4 q .. + 3] at (64 t + lane) * 4 -- a linear, conflict-free ds_read_b128
constexpr size_t FRAG_BLK = 320;                       // floats per (type, tile)
constexpr size_t FRAG = (size_t)NT * 8 * FRAG_BLK;     // 10240
constexpr size_t A_IMG = A_BBV + H;
constexpr size_t IMG_FRAG_K = 0;
constexpr size_t IMG_FRAG_V = IMG_FRAG_K + FRAG;
constexpr size_t IMG_WT = IMG_FRAG_V + FRAG;
constexpr size_t IMG_LN = IMG_WT + NT * 2 * H;
constexpr size_t IMG_WBV = IMG_LN + 4 * H;
constexpr size_t IMG_SIZE_H2X = IMG_WBV + (size_t)HEADS * H;  // 24064 floats = 96256 B
constexpr size_t IMG_SIZE_X2H = IMG_WBV + (size_t)H * H; // 38400 floats = 153600 B
// fragment-ordered tables of the MFMA node kernels (node_mfma.hip)
constexpr size_t A_NPROJ_FRAG = A_IMG + IMG_SIZE_X2H;              // [10 ch][4 ct][8 s4][64 lanes][4]
constexpr size_t A_WQ1_FRAG = A_NPROJ_FRAG + (size_t)H * PROW;     // [8 nt][8 s4][64][4]
constexpr size_t A_WBK_FRAG = A_WQ1_FRAG + (size_t)H * H;          // [16 a][2 g][64][8], pre-scaled by 1/sqrt(8)
// centred first Linears (W - colmean(W), b - mean(b)) of k and v, reference layout [128][340]: the source of
// every fragment table above; and the per-destination-class bias rows of the MFMA node projection
//   bn2[lig_i][640] = [bkc + Wkc[:, type(src prot, lig_i)] | bvc + Wvc[:, type(src prot, lig_i)] | 0 | 0 | bq0]
constexpr size_t A_WAKC = A_WBK_FRAG + (size_t)H * H;
constexpr size_t A_BAKC = A_WAKC + (size_t)H * KV_IN;
constexpr size_t A_WAVC = A_BAKC + H;
constexpr size_t A_BAVC = A_WAVC + (size_t)H * KV_IN;
constexpr size_t A_BN2 = A_BAVC + H;
// backward-only copies: Wbk transposed [m][n] and the second q Linear in the reference layout [n][k]
constexpr size_t A_WBKT = A_BN2 + 2 * PROW;
constexpr size_t A_WQ1O = A_WBKT + (size_t)H * H;
// rbf columns of the first Linears, transposed and padded for the backward's B operand: [4 types][256 c][32 g]
constexpr size_t A_WRT = A_WQ1O + (size_t)H * H;
// rbf columns of the *centred* first Linears, same layout as A_WR: the backward's recomputation when the node projection
// comes from the MFMA node kernel (whose P is centred and already contains the protein-source type column)
constexpr size_t A_WRC = A_WRT + (size_t)NT * 2 * H * 32;
// training only (x2h blocks): the split-f16 rbf table of the v path in the EDGE-major labeling (A operand), as IMG_FRAG_K
// is for k -- the x2h forward image keeps v channel-major.  Read from L2 by train_bwd_x2h.hip.
constexpr size_t A_FRAGV_EM = A_WRC + (size_t)NT * G * 2 * H;
// ---- range-safe split-f16: every f16 table is scaled by exact powers of two at pack time so that its
// largest entry sits in [2^14, 2^15) -- hi = f16(w 2^k), lo = f16(w 2^k - hi) then keep 2^-21 relative accuracy over 18 binades
// below the table's maximum instead of losing `lo` to f16 subnormals below |w| = 2^-3 -- and the scale is undone in fp32.
//   A_RBF_SC   [2 paths k|v][4]  {S = 2^(kw + RBF_EXP), 1 / (H S^2), 1 / S, kw}: the rbf tables of a path share one exponent
//                                kw (both source classes accumulate into one tile); the kernels scale the rbf values by
//                                2^RBF_EXP and PD + PS by S, LayerNorm divides it out again
//   A_NPROJ_CINV [640]           2^-kc per output column of the node projection tables (A_NPROJ_FRAG holds Wn[:, col] 2^kc)
//   A_WQ1_CINV   [128]           same for the query MLP's second Linear (A_WQ1_FRAG)
// The activations (rows of h, LayerNorm outputs) are scaled per row inside the node kernels.
constexpr int RBF_EXP = 12;          // rbf values in [0, 1] -> [0, 2^12]
constexpr int RBF_KW_MIN = -40;      // S >= 2^-28: 1 / (H S^2) stays a finite fp32 number (weights up to 2^54 keep full precision)
constexpr int RBF_KW_MAX = 20;       // S <= 2^32: sum of squares of 128 scaled pre-activations stays finite up to |pre| = 2^28
constexpr size_t A_RBF_SC = A_FRAGV_EM + FRAG;
constexpr size_t A_NPROJ_CINV = A_RBF_SC + 8;
constexpr size_t A_WQ1_CINV = A_NPROJ_CINV + PROW;
// ---- x2h blocks: LDS image of the edge kernel's PROTEIN-ONLY specialisation (edge_mfma.hip, PP = true): destinations whose 32
// neighbours and themselves are protein atoms (~ 3 of 4 nodes of a pocket) need the rbf tables of edge type 3 only, which leaves
// room for the second k Linear next to the second v Linear -- the query is then folded in registers and Qt[N,16,128] (8 KB
// written + 8 KB read per node and block) never exists for those nodes.  One contiguous region, copied verbatim into LDS:
//   frag_k [8 t][320]   type-3 slice of IMG_FRAG_K          frag_v [8 t][320]   type-3 slice of IMG_FRAG_V
//   ln [4][128]         as IMG_LN                           wbv [128][128]      as IMG_WBV (swizzled chunks)
//   wfold [8 t][8 d][64 lanes][4]   Wbk[8c + d][16t + 4q + r] / sqrt(8) for lane (c, q): lane (c = head, q) folds
//                       Qt[c][16t + 4q + r] = sum_d q[8c + d] wfold[t][d][lane][r] with conflict-free linear ds_read_b128
constexpr size_t PP_FRAG_K = 0;
constexpr size_t PP_FRAG_V = PP_FRAG_K + 8 * FRAG_BLK;
constexpr size_t PP_LN = PP_FRAG_V + 8 * FRAG_BLK;
constexpr size_t PP_WBV = PP_LN + 4 * H;
constexpr size_t PP_WFOLD = PP_WBV + (size_t)H * H;
constexpr size_t PP_IMG_SIZE = PP_WFOLD + (size_t)H * H;      // 38400 floats = 153600 B, the size of the general image
constexpr size_t A_IMG_PP = A_WQ1_CINV + H;
constexpr size_t ATT_SIZE = A_IMG_PP + PP_IMG_SIZE;

constexpr size_t LAYER_SIZE = 2 * ATT_SIZE;       // x2h then h2x

// ---- classifier ----
constexpr size_t C_W0T = 0;                       // [128][128] K-major
constexpr size_t C_B0 = C_W0T + (size_t)H * H;    // [128]
constexpr size_t C_W1T = C_B0 + H;                // [128][C] K-major

inline size_t cls_size(int C) { return C_W1T + (size_t)H * C + ((C + 3) / 4) * 4; }
inline size_t cls_b1(int C) { return C_W1T + (size_t)H * C; }
inline size_t layer_off(int l) { return GATE_SIZE + (size_t)l * LAYER_SIZE; }
inline size_t x2h_off(int l) { return layer_off(l); }
inline size_t h2x_off(int l) { return layer_off(l) + ATT_SIZE; }
inline size_t cls_off(int L) { return GATE_SIZE + (size_t)L * LAYER_SIZE; }
inline size_t packed_floats(int L, int C) { return cls_off(L) + cls_size(C); }

}  // namespace cbgx
