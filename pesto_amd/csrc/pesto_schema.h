// pesto_schema.h - weight-blob schema (host blob order) and the re-laid-out device image.
//
// Host blob order = reference state_dict order minus m_nn/sdk (pesto_amd/weights.py is the Python
// statement of the same table; reference model/model.py:10-30, src/model_operations.py:27-85,172-195).
// Device image: every Linear stored TRANSPOSED, Wt[in][out] (lanes index `out`, so a wave reads one k-row
// coalesced), with the three edge MLPs' same-depth layers concatenated along `out`:
//   edge layer 1: Wt1[193][128]  cols 0-31 eqkm, 32-63 epkm, 64-127 evm ; b1[128]
//   edge layer 2: three blocks   eq Wt[32][32], ep Wt[32][32], ev Wt[64][64] ; b2[128]
//   edge layer 3: eq Wt[32][3], ep Wt[32][9], ev Wt[64][64] ; b3[76]  (cols 0-2 Kq, 3-11 Kp, 12-75 V)
#pragma once
#include <cstdint>
#include <vector>

#include "../../include/pesto_hip.h"

namespace pesto {

constexpr int S = 32;    // Ns
constexpr int NH = 2;    // attention heads
constexpr int NK = 3;    // key size
constexpr int PH = 4;    // pool heads
constexpr int XE = 6 * S + 1;  // 193
constexpr int KMAX = PESTO_MAX_K;

// offsets (in floats) into the device weight image
struct LinearW { int32_t w, b, n_in, n_out; };  // b < 0: no bias
struct MlpW { LinearW l[3]; int32_t depth; };
struct LayerW {
    int32_t w1, b1;             // [193][128], [128]
    int32_t w2eq, w2ep, w2ev;   // [32][32], [32][32], [64][64]
    int32_t b2;                 // [128]
    int32_t w3eq, w3ep, w3ev;   // [32][3], [32][9], [64][64]
    int32_t b3;                 // [76]
    MlpW nqm, qpm;
    LinearW ppm;
    int32_t nn;
    // ---- MFMA path (pesto_node.hip, pesto_edge.hip). "frag" tables are MFMA operand fragments in the order
    // [out-block m][in-block fb][lane 0..63][r 0..3]: value W[16m + (lane&15)][16fb + 4(lane>>4) + r], so one
    // float4 per lane feeds the four k-steps r of mfma_f32_16x16x4 for (m, fb).
    int32_t e_lds;              // edge-kernel LDS image, EDGE_LDS_FLOATS contiguous floats (layout: EdgeLds below)
    int32_t e_lds16;            // same image with the value network's layer-2/3 fragments as f16 hi/lo pairs for
                                // v_mfma_f32_16x16x32_f16: [m][kgroup][hi|lo][lane][8 halves], k(kg, j) = 16(2 kgroup + j/4) + 4kg + j%4
    int32_t n_q0, n_bq0, n_q1, n_bq1, n_q2, n_bq2, n_pp;   // qpm frags [2][4],[2][2],[2][2] + biases[32]; ppm frags [2][4]
    int32_t n_ua, n_b1, n_gc;                               // [U|A] frags [16][4], b1[128]; [G|C] frags [16][2]
    int32_t n_b1s;                                          // b1 x log2(e): bias of U on the f16-split path (log2-domain ELU, pesto_schema.cpp)
    int32_t n_bn2s;                                         // nqm's last bias x log2(e) / sdk: the split path's queries carry the softmax scale (pesto_schema.cpp)
    int32_t n_n0, n_bn0, n_n1, n_bn1, n_n2, n_bn2;          // nqm frags [2][4],[2][2],[1][2] + biases (last padded to 16)
    // f16 hi/lo fragment tables of the same matrices (layout of put_frags_f16: [m][kgroup][hi|lo][lane][8 halves])
    int32_t h_q0, h_q1, h_q2, h_pp, h_ua, h_gc, h_n0, h_n1, h_n2;
};

// edge-kernel LDS image (float offsets)
constexpr int EL_W2F = 0;                 // 24 frags: eq (m,fb) 2x2, ep 2x2, ev 4x4
constexpr int EL_W3K = EL_W2F + 24 * 256; // 4 frags: key rows arranged [part g][kappa], K = h2 blocks 0..3
constexpr int EL_W3V = EL_W3K + 4 * 256;  // 16 frags [fo][m]: value layer 3, B-operand orientation
constexpr int EL_B2 = EL_W3V + 16 * 256;  // 128
constexpr int EL_BK = EL_B2 + 128;        // 16
constexpr int EL_B3V = EL_BK + 16;        // 64
constexpr int EL_WD = EL_B3V + 64;        // 128  (column 0 of edge layer 1: the distance weight)
constexpr int EDGE_LDS_FLOATS = EL_WD + 128;   // 11600
// hybrid first layer (e_lds16 image only): f16 hi/lo fragments of W1[:, 161:193] (the p_j.r block), linear-k layout
// [mb 8][hi|lo][lane][8 halves] (put_frags_f16_linear)
constexpr int EL_W1P = EDGE_LDS_FLOATS;
constexpr int EDGE_LDS_FLOATS_HY = EL_W1P + 8 * 2 * 256;   // 15696

// per-atom records written by the node kernel and read by the edge kernel (float counts)
constexpr int REC_NB = 512;    // [fb 8][g 4][A,C0,C1,C2][r 4]  (p_j itself is gathered from the state array)
constexpr int REC_A = 128;     // hybrid path: only A_j[f] (natural feature order); the C_j terms are recomputed per edge from p_j
constexpr int REC_CEN = 528;   // [fb 8][kg 4: G0,G1,G2,U][f 16] = 512, then Q[12] padded to 16
constexpr int REC_Z = 256;     // Zq[h*32+s] (64), Zp[c][h*32+s] (192)
struct ModelW {
    MlpW em, sam, zdm, dm;
    LinearW zdm_vec;
};

struct HostLinear { int64_t w, b; int n_in, n_out; };
struct HostMlp { HostLinear l[3]; int depth; };
struct HostLayer { HostMlp nqm, eqkm, epkm, evm, qpm; HostLinear ppm; };
struct HostSchema {
    HostMlp em, sam, zdm, dm;
    HostLinear zdm_vec;
    std::vector<HostLayer> layers;
    int64_t total = 0;
};

// ---- host-side packing helpers of pesto_forward_batch_submit (plain C++: pesto_schema.cpp is compiled for the host only)
// one-hot rows -> byte indices: every block [bounds[c], bounds[c + 1]) of every row of q [n, n0] must hold exactly one 1.0f among +0.0f
// (encode_features, src/data_encoding.py:78-84) and the columns in front of bounds[0] only zeros; false at the first row that does not
bool onehot_rows_to_indices(const float* q, int64_t n, int n0, const int* bounds, int n_index, uint8_t* dst);
// neighbour ids -> uint16 (kind: 32 = int32, 64 = int64): false when an entry is outside [0, 65535] (narrowing must not wrap an invalid id into a valid one)
bool narrow_ids_to_u16(const void* src, int kind, size_t count, uint16_t* dst);

bool config_ok(const pesto_config* c);
HostSchema host_schema(const pesto_config& c);

// builds the device image (host-side vector) + offset tables from the host blob
struct DeviceImage {
    std::vector<float> data;
    ModelW model;
    std::vector<LayerW> layers;
};
DeviceImage build_device_image(const pesto_config& c, const float* blob);

}  // namespace pesto
