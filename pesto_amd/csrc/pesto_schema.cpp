// pesto_schema.cpp - host blob schema and device weight image construction (host code only).
#include "pesto_schema.h"

#include <cmath>
#include <cstring>

namespace pesto {

bool onehot_rows_to_indices(const float* q, int64_t n, int n0, const int* bounds, int n_index, uint8_t* dst) {
    // branch-free per block so that the compiler vectorises the scans (the rows are read as words: 1.0f = 0x3f800000, +0.0f = 0):
    // ones = words equal to 1.0f, other = words that are neither 0 nor 1.0f, at = sum of (f - begin) over the ones
    constexpr uint32_t ONE = 0x3f800000u;
    for (int64_t i = 0; i < n; ++i) {
        const uint32_t* row = reinterpret_cast<const uint32_t*>(q) + i * n0;
        uint32_t lead = 0;
        for (int f = 0; f < bounds[0]; ++f) lead |= row[f];
        if (lead) return false;
        for (int c = 0; c < n_index; ++c) {
            const int b0 = bounds[c], b1 = bounds[c + 1];
            int ones = 0, other = 0, at = 0;
            for (int f = b0; f < b1; ++f) {
                const int is1 = row[f] == ONE ? 1 : 0;
                ones += is1;
                other += (row[f] != 0u && !is1) ? 1 : 0;
                at += is1 * (f - b0);
            }
            if (ones != 1 || other != 0 || at > 255) return false;
            dst[i * n_index + c] = (uint8_t)at;
        }
    }
    return true;
}

bool narrow_ids_to_u16(const void* src, int kind, size_t count, uint16_t* dst) {
    if (kind == 64) {
        const int64_t* p = static_cast<const int64_t*>(src);
        uint64_t bad = 0;
        for (size_t i = 0; i < count; ++i) { bad |= (uint64_t)p[i]; dst[i] = (uint16_t)p[i]; }      // (negative ids set the high bits too)
        return (bad >> 16) == 0;
    }
    const int32_t* p = static_cast<const int32_t*>(src);
    uint32_t bad = 0;
    for (size_t i = 0; i < count; ++i) { bad |= (uint32_t)p[i]; dst[i] = (uint16_t)p[i]; }
    return (bad >> 16) == 0;
}

bool config_ok(const pesto_config* c) {
    if (!c || c->n0 < 1 || c->n0 > 512 || c->n_layers < 1 || c->n_layers > PESTO_MAX_LAYERS) return false;
    if (c->n_out < 1 || c->n_out > 32) return false;
    if ((c->em_depth != 1 && c->em_depth != 3) || (c->dm_depth != 1 && c->dm_depth != 3)) return false;
    if (c->precision != PESTO_PRECISION_AUTO && c->precision != PESTO_PRECISION_F16_SPLIT && c->precision != PESTO_PRECISION_FP32) return false;
    for (int l = 0; l < c->n_layers; ++l)
        if (c->nn[l] != 8 && c->nn[l] != 16 && c->nn[l] != 32 && c->nn[l] != 64) return false;
    return true;
}

namespace {
struct Cursor {
    int64_t off = 0;
    HostLinear lin(int n_in, int n_out, bool bias) {
        HostLinear l{off, -1, n_in, n_out};
        off += (int64_t)n_in * n_out;
        if (bias) { l.b = off; off += n_out; }
        return l;
    }
    HostMlp mlp3(int d0, int d1, int d2, int d3) {
        HostMlp m; m.depth = 3;
        m.l[0] = lin(d0, d1, true); m.l[1] = lin(d1, d2, true); m.l[2] = lin(d2, d3, true);
        return m;
    }
    HostMlp mlp1(int d0, int d1) {
        HostMlp m{}; m.depth = 1;
        m.l[0] = lin(d0, d1, true);
        return m;
    }
};

// appends Wt[in][out] (transpose of torch's W[out][in]) to the image, returns its offset
int32_t put_transposed(std::vector<float>& img, const float* blob, const HostLinear& l) {
    int32_t off = (int32_t)img.size();
    img.resize(img.size() + (size_t)l.n_in * l.n_out);
    for (int o = 0; o < l.n_out; ++o)
        for (int i = 0; i < l.n_in; ++i) img[off + (size_t)i * l.n_out + o] = blob[l.w + (int64_t)o * l.n_in + i];
    return off;
}
int32_t put_bias(std::vector<float>& img, const float* blob, const HostLinear& l) {
    if (l.b < 0) return -1;
    int32_t off = (int32_t)img.size();
    img.insert(img.end(), blob + l.b, blob + l.b + l.n_out);
    return off;
}
LinearW put_linear(std::vector<float>& img, const float* blob, const HostLinear& l) {
    LinearW d;
    d.w = put_transposed(img, blob, l);
    d.b = put_bias(img, blob, l);
    d.n_in = l.n_in; d.n_out = l.n_out;
    return d;
}
MlpW put_mlp(std::vector<float>& img, const float* blob, const HostMlp& m) {
    MlpW d{}; d.depth = m.depth;
    for (int i = 0; i < m.depth; ++i) d.l[i] = put_linear(img, blob, m.l[i]);
    return d;
}
void pad16(std::vector<float>& img) { while (img.size() % 4) img.push_back(0.0f); }

// A dense row-major matrix view W[rows][cols] assembled from pieces of the blob (zero outside)
struct Mat {
    int rows, cols;
    std::vector<float> v;
    Mat(int r, int c) : rows(r), cols(c), v((size_t)r * c, 0.0f) {}
    float& at(int r, int c) { return v[(size_t)r * cols + c]; }
    float get(int r, int c) const { return (r < rows && c < cols) ? v[(size_t)r * cols + c] : 0.0f; }
    Mat scaled(float f) const { Mat m = *this; for (float& x : m.v) x *= f; return m; }
};
// The f16-split kernels evaluate ELU in the log2 domain: pre-activations arrive multiplied by log2(e), so that exp(x) is a
// bare v_exp_f32 (2^t) without the scaling multiply, and activations leave multiplied by log2(e) as well:
//   t = c x,  c ELU(x) = med3(t, c 2^t - c, 0)   (c = log2 e)
// Per edge MLP  x -> L1 -> ELU -> L2 -> ELU -> L3 : every term of the first pre-activation is scaled by c (U, G, A records,
// the p_j.r block, the distance column, b1), W2 is unchanged (its input already carries the c) with b2 scaled by c, W3 is divided
// by c. Only the tables the split kernels read are touched; the exact fp32 kernels keep the reference's numbers.
constexpr float LOG2E = 1.44269504088896340736f;
// copy torch weight W[out][in] columns [c0, c0+nc) of rows [0, n_out) into M at (r0, k0)
void put_block(Mat& M, int r0, int k0, const float* blob, const HostLinear& l, int c0, int nc, int row_lo = 0, int row_n = -1) {
    if (row_n < 0) row_n = l.n_out;
    for (int o = 0; o < row_n; ++o)
        for (int c = 0; c < nc; ++c) M.at(r0 + o, k0 + c) = blob[l.w + (int64_t)(row_lo + o) * l.n_in + c0 + c];
}
// MFMA fragment table [m][fb][lane][r] of M (rows = out, cols = in), appended to img; returns offset
int32_t put_frags(std::vector<float>& img, const Mat& M, int n_m, int n_fb, int row0 = 0, int col0 = 0) {
    int32_t off = (int32_t)img.size();
    for (int m = 0; m < n_m; ++m)
        for (int fb = 0; fb < n_fb; ++fb)
            for (int lane = 0; lane < 64; ++lane)
                for (int r = 0; r < 4; ++r)
                    img.push_back(M.get(row0 + 16 * m + (lane & 15), col0 + 16 * fb + 4 * (lane >> 4) + r));
    return off;
}
// f16 hi/lo fragment table for v_mfma_f32_16x16x32_f16 over the image (two halves per float slot):
// [m][kgroup][hi|lo][lane][j 0..7] = split(W[16m + (lane&15)][16(2 kgroup + j/4) + 4(lane>>4) + j%4]), w = hi + lo
// with hi = f16(w) (round to nearest), lo = f16(w - hi): w is represented to ~2^-22 relative.
int32_t put_frags_f16(std::vector<float>& img, const Mat& M, int n_m, int n_kg) {
    std::vector<_Float16> h;
    for (int m = 0; m < n_m; ++m)
        for (int kgp = 0; kgp < n_kg; ++kgp)
            for (int part = 0; part < 2; ++part)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const float w = M.get(16 * m + (lane & 15), 16 * (2 * kgp + (j >> 2)) + 4 * (lane >> 4) + (j & 3));
                        const _Float16 hi = (_Float16)w;
                        h.push_back(part == 0 ? hi : (_Float16)(w - (float)hi));
                    }
    const size_t off = img.size();
    img.resize(off + h.size() / 2);
    memcpy(&img[off], h.data(), h.size() * sizeof(_Float16));
    return (int32_t)off;
}
// same hi/lo tables for an operand whose k index is NOT a chained MFMA output but data gathered by the kernel itself: lane
// (row, kg) feeds k = 32 kgroup + 8 kg + j, i.e. eight CONSECUTIVE input features (one 32-byte gather per lane):
// [m][kgroup][hi|lo][lane][j] = split(W[16m + (lane&15)][32 kgroup + 8(lane>>4) + j])
int32_t put_frags_f16_linear(std::vector<float>& img, const Mat& M, int n_m, int n_kg) {
    std::vector<_Float16> h;
    for (int m = 0; m < n_m; ++m)
        for (int kgp = 0; kgp < n_kg; ++kgp)
            for (int part = 0; part < 2; ++part)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const float w = M.get(16 * m + (lane & 15), 32 * kgp + 8 * (lane >> 4) + j);
                        const _Float16 hi = (_Float16)w;
                        h.push_back(part == 0 ? hi : (_Float16)(w - (float)hi));
                    }
    const size_t off = img.size();
    img.resize(off + h.size() / 2);
    memcpy(&img[off], h.data(), h.size() * sizeof(_Float16));
    return (int32_t)off;
}
int32_t put_vec(std::vector<float>& img, const float* src, int n, int pad_to = 0) {
    int32_t off = (int32_t)img.size();
    img.insert(img.end(), src, src + n);
    for (int i = n; i < pad_to; ++i) img.push_back(0.0f);
    return off;
}
}  // namespace

HostSchema host_schema(const pesto_config& c) {
    HostSchema h;
    Cursor cur;
    h.em = c.em_depth == 3 ? cur.mlp3(c.n0, S, S, S) : cur.mlp1(c.n0, S);
    h.layers.resize(c.n_layers);
    for (int l = 0; l < c.n_layers; ++l) {
        HostLayer& L = h.layers[l];
        L.nqm = cur.mlp3(2 * S, S, S, 2 * NK * NH);
        L.eqkm = cur.mlp3(XE, S, S, NK);
        L.epkm = cur.mlp3(XE, S, S, 3 * NK);
        L.evm = cur.mlp3(XE, 2 * S, 2 * S, 2 * S);
        L.qpm = cur.mlp3(NH * S, S, S, S);
        L.ppm = cur.lin(NH * S, S, false);
    }
    h.sam = cur.mlp3(2 * S, S, S, 2 * PH);
    h.zdm = cur.mlp3(PH * S, S, S, S);
    h.zdm_vec = cur.lin(PH * S, S, false);
    h.dm = c.dm_depth == 3 ? cur.mlp3(2 * S, S, S, c.n_out) : cur.mlp1(2 * S, c.n_out);
    h.total = cur.off;
    return h;
}

DeviceImage build_device_image(const pesto_config& c, const float* blob) {
    HostSchema h = host_schema(c);
    DeviceImage d;
    std::vector<float>& img = d.data;
    d.model.em = put_mlp(img, blob, h.em);
    d.model.sam = put_mlp(img, blob, h.sam);
    d.model.zdm = put_mlp(img, blob, h.zdm);
    d.model.zdm_vec = put_linear(img, blob, h.zdm_vec);
    d.model.dm = put_mlp(img, blob, h.dm);
    d.layers.resize(c.n_layers);
    for (int l = 0; l < c.n_layers; ++l) {
        const HostLayer& L = h.layers[l];
        LayerW& W = d.layers[l];
        W.nn = c.nn[l];
        pad16(img);
        // edge layer 1: concatenated along out -> Wt1[193][128]
        W.w1 = (int32_t)img.size();
        img.resize(img.size() + (size_t)XE * 128);
        const HostMlp* nets[3] = {&L.eqkm, &L.epkm, &L.evm};
        const int col0[3] = {0, 32, 64};
        for (int n = 0; n < 3; ++n) {
            const HostLinear& l0 = nets[n]->l[0];
            for (int o = 0; o < l0.n_out; ++o)
                for (int i = 0; i < XE; ++i) img[W.w1 + (size_t)i * 128 + col0[n] + o] = blob[l0.w + (int64_t)o * XE + i];
        }
        W.b1 = (int32_t)img.size();
        for (int n = 0; n < 3; ++n) img.insert(img.end(), blob + nets[n]->l[0].b, blob + nets[n]->l[0].b + nets[n]->l[0].n_out);
        W.w2eq = put_transposed(img, blob, L.eqkm.l[1]);
        W.w2ep = put_transposed(img, blob, L.epkm.l[1]);
        W.w2ev = put_transposed(img, blob, L.evm.l[1]);
        W.b2 = (int32_t)img.size();
        for (int n = 0; n < 3; ++n) img.insert(img.end(), blob + nets[n]->l[1].b, blob + nets[n]->l[1].b + nets[n]->l[1].n_out);
        W.w3eq = put_transposed(img, blob, L.eqkm.l[2]);
        W.w3ep = put_transposed(img, blob, L.epkm.l[2]);
        W.w3ev = put_transposed(img, blob, L.evm.l[2]);
        W.b3 = (int32_t)img.size();
        for (int n = 0; n < 3; ++n) img.insert(img.end(), blob + nets[n]->l[2].b, blob + nets[n]->l[2].b + nets[n]->l[2].n_out);
        W.nqm = put_mlp(img, blob, L.nqm);
        W.qpm = put_mlp(img, blob, L.qpm);
        W.ppm = put_linear(img, blob, L.ppm);

        // ---------------- MFMA path images
        // W1cat[128][193]: rows 0-31 eqkm, 32-63 epkm, 64-127 evm (first Linear of each edge MLP)
        Mat W1(128, XE);
        std::vector<float> b1(128), b2(128);
        for (int n = 0; n < 3; ++n) {
            put_block(W1, col0[n], 0, blob, nets[n]->l[0], 0, XE);
            for (int o = 0; o < nets[n]->l[0].n_out; ++o) b1[col0[n] + o] = blob[nets[n]->l[0].b + o];
            for (int o = 0; o < nets[n]->l[1].n_out; ++o) b2[col0[n] + o] = blob[nets[n]->l[1].b + o];
        }
        pad16(img);
        W.e_lds = (int32_t)img.size();
        Mat Wk16(16, 64);
        {   // edge layer 2 frags: eq 2x2, ep 2x2, ev 4x4
            Mat Weq(32, 32), Wep(32, 32), Wev(64, 64);
            put_block(Weq, 0, 0, blob, L.eqkm.l[1], 0, 32);
            put_block(Wep, 0, 0, blob, L.epkm.l[1], 0, 32);
            put_block(Wev, 0, 0, blob, L.evm.l[1], 0, 64);
            put_frags(img, Weq, 2, 2); put_frags(img, Wep, 2, 2); put_frags(img, Wev, 4, 4);
            // key rows: row 4*part + kappa; part 0 = eqkm out kappa (reads h2[0:32]); part 1..3 = epkm out 3(part-1)+kappa (h2[32:64])
            Mat Wk(16, 64);
            std::vector<float> bk(16, 0.0f);
            for (int kap = 0; kap < 3; ++kap) {
                for (int c = 0; c < 32; ++c) Wk.at(kap, c) = blob[L.eqkm.l[2].w + (int64_t)kap * 32 + c];
                bk[kap] = blob[L.eqkm.l[2].b + kap];
                for (int part = 1; part < 4; ++part) {
                    const int o = 3 * (part - 1) + kap;
                    for (int c = 0; c < 32; ++c) Wk.at(4 * part + kap, 32 + c) = blob[L.epkm.l[2].w + (int64_t)o * 32 + c];
                    bk[4 * part + kap] = blob[L.epkm.l[2].b + o];
                }
            }
            put_frags(img, Wk, 1, 4);
            Wk16 = Wk;
            Mat Wv(64, 64);
            put_block(Wv, 0, 0, blob, L.evm.l[2], 0, 64);
            put_frags(img, Wv, 4, 4);               // [fo][m]
            put_vec(img, b2.data(), 128);
            put_vec(img, bk.data(), 16);
            put_vec(img, blob + L.evm.l[2].b, 64);
            std::vector<float> wd(128);
            for (int f = 0; f < 128; ++f) wd[f] = W1.get(f, 0);
            put_vec(img, wd.data(), 128);
        }
        {   // f16-split variant of the same LDS image: only the value network's big GEMMs change representation
            pad16(img);
            W.e_lds16 = (int32_t)img.size();
            img.insert(img.end(), img.begin() + W.e_lds, img.begin() + W.e_lds + EDGE_LDS_FLOATS);
            for (int f = 0; f < 128; ++f) { img[W.e_lds16 + EL_B2 + f] *= LOG2E; img[W.e_lds16 + EL_WD + f] *= LOG2E; }    // log2-domain ELU (above)
            Mat Wev2(64, 64), Wv3(64, 64);
            put_block(Wev2, 0, 0, blob, L.evm.l[1], 0, 64);
            put_block(Wv3, 0, 0, blob, L.evm.l[2], 0, 64);
            Wv3 = Wv3.scaled(1.0f / LOG2E);
            std::vector<float> tmp;
            put_frags_f16(tmp, Wev2, 4, 2);              // 4 out-blocks x 2 k-groups x (hi, lo) x 256 floats = 4096 floats
            std::copy(tmp.begin(), tmp.end(), img.begin() + W.e_lds16 + EL_W2F + 8 * 256);
            tmp.clear();
            put_frags_f16(tmp, Wv3, 4, 2);
            std::copy(tmp.begin(), tmp.end(), img.begin() + W.e_lds16 + EL_W3V);
            // key networks: eq / ep layer 2 (K = 32: one k-group each) and the arranged key rows (K = 64: two k-groups)
            Mat Weq2(32, 32), Wep2(32, 32);
            put_block(Weq2, 0, 0, blob, L.eqkm.l[1], 0, 32);
            put_block(Wep2, 0, 0, blob, L.epkm.l[1], 0, 32);
            tmp.clear();
            put_frags_f16(tmp, Weq2, 2, 1);             // 2 x 1 x 2 x 256 = 1024 floats (same as four fp32 fragments)
            put_frags_f16(tmp, Wep2, 2, 1);
            std::copy(tmp.begin(), tmp.end(), img.begin() + W.e_lds16 + EL_W2F);
            tmp.clear();
            put_frags_f16(tmp, Wk16.scaled(1.0f / LOG2E), 1, 2);
            std::copy(tmp.begin(), tmp.end(), img.begin() + W.e_lds16 + EL_W3K);
            // hybrid first layer: the p_j.r block of edge layer 1 (W1[:, 161:193]) applied per edge on the matrix cores
            Mat W1p(128, 32);
            for (int f = 0; f < 128; ++f)
                for (int c = 0; c < 32; ++c) W1p.at(f, c) = W1.get(f, 161 + c);
            const int32_t off = put_frags_f16_linear(img, W1p.scaled(LOG2E), 8, 1);      // appended right behind the 11600-float image
            if (off != W.e_lds16 + EL_W1P) abort();
        }
        // node kernel: finish (qpm, ppm)
        {
            Mat M0(32, 64), M1(32, 32), M2(32, 32), Mp(32, 64);
            put_block(M0, 0, 0, blob, L.qpm.l[0], 0, 64);
            put_block(M1, 0, 0, blob, L.qpm.l[1], 0, 32);
            put_block(M2, 0, 0, blob, L.qpm.l[2], 0, 32);
            put_block(Mp, 0, 0, blob, L.ppm, 0, 64);
            W.h_q0 = put_frags_f16(img, M0, 2, 2); W.h_q1 = put_frags_f16(img, M1, 2, 1); W.h_q2 = put_frags_f16(img, M2, 2, 1);
            W.h_pp = put_frags_f16(img, Mp, 2, 2);
            W.n_q0 = put_frags(img, M0, 2, 4); W.n_bq0 = put_vec(img, blob + L.qpm.l[0].b, 32);
            W.n_q1 = put_frags(img, M1, 2, 2); W.n_bq1 = put_vec(img, blob + L.qpm.l[1].b, 32);
            W.n_q2 = put_frags(img, M2, 2, 2); W.n_bq2 = put_vec(img, blob + L.qpm.l[2].b, 32);
            W.n_pp = put_frags(img, Mp, 2, 4);
        }
        // node kernel: records. [U|A]: rows 0-127 = W1[:, 1:65] (centre role), rows 128-255 = W1[:, 65:129] (neighbour role)
        {
            Mat Mua(256, 64), Mgc(256, 32);
            for (int f = 0; f < 128; ++f)
                for (int c = 0; c < 64; ++c) { Mua.at(f, c) = W1.get(f, 1 + c); Mua.at(128 + f, c) = W1.get(f, 65 + c); }
            for (int f = 0; f < 128; ++f)
                for (int c = 0; c < 32; ++c) { Mgc.at(f, c) = W1.get(f, 129 + c); Mgc.at(128 + f, c) = W1.get(f, 161 + c); }
            W.h_ua = put_frags_f16(img, Mua.scaled(LOG2E), 16, 2);       // records of the split path: log2-domain (above)
            W.h_gc = put_frags_f16(img, Mgc.scaled(LOG2E), 16, 1);
            std::vector<float> b1s(b1);
            for (float& x : b1s) x *= LOG2E;
            W.n_b1s = put_vec(img, b1s.data(), 128);
            W.n_ua = put_frags(img, Mua, 16, 4);
            W.n_b1 = put_vec(img, b1.data(), 128);
            W.n_gc = put_frags(img, Mgc, 16, 2);
            Mat N0(32, 64), N1(32, 32), N2(16, 32);
            put_block(N0, 0, 0, blob, L.nqm.l[0], 0, 64);
            put_block(N1, 0, 0, blob, L.nqm.l[1], 0, 32);
            put_block(N2, 0, 0, blob, L.nqm.l[2], 0, 32);
            // The split path's node queries carry the softmax scale: Q' = Q log2(e) / sdk (sdk = sqrt(Nk), model_operations.py:85,139-140), so
            // a logit Q'.K arrives as t = log2(e) x and the kernel's softmax is exp2(t) / sum exp2(t) - no per-edge multiply, a bare v_exp_f32.
            const float QS = LOG2E * (1.0f / std::sqrt((float)NK));
            W.h_n0 = put_frags_f16(img, N0, 2, 2); W.h_n1 = put_frags_f16(img, N1, 2, 1); W.h_n2 = put_frags_f16(img, N2.scaled(QS), 1, 1);
            {
                std::vector<float> bn2s(16, 0.0f);
                for (int r = 0; r < 12; ++r) bn2s[r] = blob[L.nqm.l[2].b + r] * QS;
                W.n_bn2s = put_vec(img, bn2s.data(), 16);
            }
            W.n_n0 = put_frags(img, N0, 2, 4); W.n_bn0 = put_vec(img, blob + L.nqm.l[0].b, 32);
            W.n_n1 = put_frags(img, N1, 2, 2); W.n_bn1 = put_vec(img, blob + L.nqm.l[1].b, 32);
            W.n_n2 = put_frags(img, N2, 1, 2); W.n_bn2 = put_vec(img, blob + L.nqm.l[2].b, 12, 16);
        }
    }
    pad16(img);
    return d;
}

}  // namespace pesto
