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
    // ---- v2 (MFMA) images, filled by build_device_image_v2
    int32_t v2_base;
};
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
