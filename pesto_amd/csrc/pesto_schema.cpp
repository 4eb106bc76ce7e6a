// pesto_schema.cpp - host blob schema and device weight image construction (host code only).
#include "pesto_schema.h"

namespace pesto {

bool config_ok(const pesto_config* c) {
    if (!c || c->n0 < 1 || c->n0 > 512 || c->n_layers < 1 || c->n_layers > PESTO_MAX_LAYERS) return false;
    if (c->n_out < 1 || c->n_out > 32) return false;
    if ((c->em_depth != 1 && c->em_depth != 3) || (c->dm_depth != 1 && c->dm_depth != 3)) return false;
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
        W.v2_base = -1;
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
    }
    pad16(img);
    return d;
}

}  // namespace pesto
