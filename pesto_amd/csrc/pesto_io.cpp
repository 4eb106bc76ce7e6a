// pesto_io.cpp - libpesto_io.so: native PDB read / clean / encode / write around the forward pass (include/pesto_io.h).
// Host-only C++ (no HIP). Behaviour restated from the reference's Python (file:line relative to /root/reference):
//   read_pdb                         src/structure_io.py:6-55  - on top of gemmi.read_pdb(max_line_length=80); gemmi is a
//                                    third-party dependency absent from the reference tree (requirements: "gemmi", unpinned),
//                                    so its documented PDB-reading behaviour is restated here (see parse_pdb) and pinned by
//                                    the reference's examples/*.pdb -> *_i0.pdb pairs (tests/golden/pdb/)
//   clean_structure ... concatenate_chains   src/structure.py:14-146
//   encode_structure / encode_features       src/data_encoding.py:54-84
//   encode_bfactor, save_pdb                 src/structure.py:185-223, src/structure_io.py:96-123
#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <set>
#include <string>
#include <unordered_set>
#include <vector>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

#include "../../include/pesto_io.h"

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
    char buf[600];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

// ---- row sums for the dense-mask reduction (pesto_io_mask_to_segments_any): plain C++ and an AVX2 twin chosen at run time (the library is
// built in one container and runs on another host: no -march flags)
bool have_avx2() {
    static const bool v = __builtin_cpu_supports("avx2");
    return v;
}
uint64_t row_sum_u8(const uint8_t* row, int64_t n) {
    uint64_t s = 0;
    for (int64_t r = 0; r < n; ++r) s += row[r];
    return s;
}
uint64_t row_sum_u32(const uint32_t* row, int64_t n) {
    uint64_t s = 0;
    for (int64_t r = 0; r < n; ++r) s += row[r];
    return s;
}
#if defined(__x86_64__)
__attribute__((target("avx2"))) uint64_t row_sum_u8_avx2(const uint8_t* row, int64_t n) {
    __m256i acc = _mm256_setzero_si256();
    const __m256i zero = _mm256_setzero_si256();
    int64_t r = 0;
    for (; r + 32 <= n; r += 32) acc = _mm256_add_epi64(acc, _mm256_sad_epu8(_mm256_loadu_si256((const __m256i*)(row + r)), zero));
    uint64_t lanes[4];
    _mm256_storeu_si256((__m256i*)lanes, acc);
    uint64_t s = lanes[0] + lanes[1] + lanes[2] + lanes[3];
    for (; r < n; ++r) s += row[r];
    return s;
}
__attribute__((target("avx2"))) uint64_t row_sum_u32_avx2(const uint32_t* row, int64_t n) {
    // 32-bit lanes cannot overflow into a false 0x3f800000 unnoticed: a second accumulator ORs the words, and a valid row's OR equals its sum
    __m256i acc = _mm256_setzero_si256(), any = _mm256_setzero_si256();
    int64_t r = 0;
    for (; r + 8 <= n; r += 8) {
        const __m256i v = _mm256_loadu_si256((const __m256i*)(row + r));
        acc = _mm256_add_epi32(acc, v);
        any = _mm256_or_si256(any, v);
    }
    uint32_t a[8], o[8];
    _mm256_storeu_si256((__m256i*)a, acc); _mm256_storeu_si256((__m256i*)o, any);
    uint64_t s = 0; uint32_t orv = 0;
    for (int k = 0; k < 8; ++k) { s += a[k]; orv |= o[k]; }
    for (; r < n; ++r) { s += row[r]; orv |= row[r]; }
    return (s == 0x3f800000u && orv == 0x3f800000u) ? s : ~0ull;
}
#else
uint64_t row_sum_u8_avx2(const uint8_t* row, int64_t n) { return row_sum_u8(row, n); }
uint64_t row_sum_u32_avx2(const uint32_t* row, int64_t n) { return row_sum_u32(row, n); }
#endif

// the reference's dict of per-atom numpy arrays
struct Atoms {
    std::vector<float> xyz;   // [N,3]
    std::vector<std::string> name, element, resname, chain, icode;
    std::vector<int64_t> resid;
    std::vector<char> het;    // 'A' (ATOM) or 'H' (HETATM)
    bool has_icode = true;    // clean_structure pops "icode"
    size_t size() const { return resid.size(); }

    void push_from(const Atoms& o, size_t i) {
        xyz.insert(xyz.end(), o.xyz.begin() + 3 * i, o.xyz.begin() + 3 * i + 3);
        name.push_back(o.name[i]); element.push_back(o.element[i]); resname.push_back(o.resname[i]);
        chain.push_back(o.chain[i]); resid.push_back(o.resid[i]); het.push_back(o.het[i]);
        if (o.has_icode) icode.push_back(o.icode[i]);
    }
    Atoms select(const std::vector<size_t>& idx) const {
        Atoms r;
        r.has_icode = has_icode;
        for (size_t i : idx) r.push_from(*this, i);
        return r;
    }
};

// ------------------------------------------------------------------------------------------------ element table
// symbols as gemmi's Element.name prints them: X (unknown), the periodic table, D (deuterium)
const char* const ELEMENTS[] = {
    "X",  "H",  "He", "Li", "Be", "B",  "C",  "N",  "O",  "F",  "Ne", "Na", "Mg", "Al", "Si", "P",  "S",  "Cl", "Ar", "K",
    "Ca", "Sc", "Ti", "V",  "Cr", "Mn", "Fe", "Co", "Ni", "Cu", "Zn", "Ga", "Ge", "As", "Se", "Br", "Kr", "Rb", "Sr", "Y",
    "Zr", "Nb", "Mo", "Tc", "Ru", "Rh", "Pd", "Ag", "Cd", "In", "Sn", "Sb", "Te", "I",  "Xe", "Cs", "Ba", "La", "Ce", "Pr",
    "Nd", "Pm", "Sm", "Eu", "Gd", "Tb", "Dy", "Ho", "Er", "Tm", "Yb", "Lu", "Hf", "Ta", "W",  "Re", "Os", "Ir", "Pt", "Au",
    "Hg", "Tl", "Pb", "Bi", "Po", "At", "Rn", "Fr", "Ra", "Ac", "Th", "Pa", "U",  "Np", "Pu", "Am", "Cm", "Bk", "Cf", "Es",
    "Fm", "Md", "No", "Lr", "Rf", "Db", "Sg", "Bh", "Hs", "Mt", "Ds", "Rg", "Cn", "Nh", "Fl", "Mc", "Lv", "Ts", "Og", "D"};
const int N_ELEMENTS = sizeof(ELEMENTS) / sizeof(ELEMENTS[0]);

const char* element_one(char c) {
    c = (char)std::toupper((unsigned char)c);
    for (int i = 1; i < N_ELEMENTS; ++i)
        if (ELEMENTS[i][1] == '\0' && ELEMENTS[i][0] == c) return ELEMENTS[i];
    return ELEMENTS[0];
}
const char* element_two(char a, char b) {
    a = (char)std::toupper((unsigned char)a);
    b = (char)std::tolower((unsigned char)b);
    for (int i = 1; i < N_ELEMENTS; ++i)
        if (ELEMENTS[i][0] == a && ELEMENTS[i][1] == b) return ELEMENTS[i];
    return ELEMENTS[0];
}
// a 2-column element field (" C", "C ", "ZN", "Se"), case-insensitive
const char* element_from_field(const char* f) {
    const bool a0 = std::isalpha((unsigned char)f[0]) != 0, a1 = std::isalpha((unsigned char)f[1]) != 0;
    if (a0 && a1) return element_two(f[0], f[1]);
    if (a0) return element_one(f[0]);
    if (a1) return element_one(f[1]);
    return ELEMENTS[0];
}
// no element columns: the PDB convention puts a one-letter element in column 14 and a two-letter one in 13-14
const char* element_from_padded_name(const char* n) {
    if (n[0] == ' ' || std::isdigit((unsigned char)n[0])) return element_one(n[1]);
    if (std::isdigit((unsigned char)n[1])) return element_one(n[0]);
    if (n[3] != ' ' && std::toupper((unsigned char)n[0]) == 'H') return ELEMENTS[1];   // "HH11"-style hydrogens
    const char* e = element_two(n[0], n[1]);
    return e != ELEMENTS[0] ? e : element_one(n[0]);
}

// ------------------------------------------------------------------------------------------------ PDB parsing
std::string field(const char* line, int len, int off, int width) {   // columns [off, off+width), blanks trimmed on both sides
    if (off >= len) return std::string();
    int a = off, b = std::min(len, off + width);
    while (a < b && std::isspace((unsigned char)line[a])) ++a;
    while (b > a && std::isspace((unsigned char)line[b - 1])) --b;
    return std::string(line + a, line + b);
}
int read_int(const char* line, int len, int off, int width) {
    int a = off, b = std::min(len, off + width);
    while (a < b && std::isspace((unsigned char)line[a])) ++a;
    int sign = 1;
    if (a < b && (line[a] == '-' || line[a] == '+')) { if (line[a] == '-') sign = -1; ++a; }
    long v = 0;
    while (a < b && std::isdigit((unsigned char)line[a])) v = v * 10 + (line[a++] - '0');
    return (int)(sign * v);
}
// residue number: decimal, or hybrid-36 once the 4 columns start with a letter (A000 = 10000)
int read_seqnum(const char* line, int len, int off) {
    if (off < len && std::isalpha((unsigned char)line[off])) {
        long v = 0;
        for (int i = off; i < std::min(len, off + 4); ++i) {
            const int c = std::toupper((unsigned char)line[i]);
            v = v * 36 + (std::isdigit(c) ? c - '0' : std::isalpha(c) ? c - 'A' + 10 : 0);
        }
        return (int)(v - 10L * 36 * 36 * 36 + 10000);
    }
    return read_int(line, len, off, 4);
}
double read_double(const char* line, int len, int off, int width) {
    char buf[32];
    int n = 0;
    for (int i = off; i < std::min(len, off + width) && n < 31; ++i) buf[n++] = line[i];
    buf[n] = '\0';
    return std::strtod(buf, nullptr);
}
bool record_is(const char* line, int len, const char* rec4) {   // first four columns, case-insensitive, blank padded
    for (int i = 0; i < 4; ++i) {
        const char c = i < len ? (char)std::toupper((unsigned char)line[i]) : ' ';
        if (c != rec4[i]) return false;
    }
    return true;
}

struct PAtom { std::string name; const char* element; char altloc; float x, y, z; };
struct PRes { int num; char icode; std::string name, segment; char het; std::vector<PAtom> atoms; };
struct PChain { std::string name; std::vector<PRes> res; };
struct PModel { std::string name; std::vector<PChain> chains; };

// What gemmi.read_pdb(path, max_line_length=80) hands to the reference's loop (src/structure_io.py:22-44):
//  * ATOM / HETATM records only, in MODEL order; a file without MODEL records is one model; reading stops at END;
//  * lines are cut at 80 columns; chain = columns 21-22, residue name = 18-20, number = 23-26 (+ insertion code 27),
//    atom name = 13-16, altloc = 17, coordinates 31-54, element from columns 77-78 when they hold a letter, else
//    inferred from the padded atom name;
//  * chain parts with the same name inside a model are merged in order of first appearance (ligands and waters listed
//    after the polymers join their chain), atoms of one residue id within a chain part are grouped;
//  * het_flag is per residue ('A' / 'H' from the record that opened it).
int parse_pdb(const char* text, int64_t n, std::vector<PModel>& models) {
    PModel* model = nullptr;
    PChain* chain = nullptr;
    PRes* res = nullptr;
    for (int64_t pos = 0; pos < n;) {
        int64_t e = pos;
        while (e < n && text[e] != '\n') ++e;
        const char* line = text + pos;
        int len = (int)std::min<int64_t>(e - pos, 80);
        pos = e + 1;
        while (len > 0 && (line[len - 1] == '\r' || line[len - 1] == '\0')) --len;
        if (len < 3) continue;
        if (record_is(line, len, "ATOM") || record_is(line, len, "HETA")) {
            if (len < 55) return fail(PESTO_IO_ERR_PARSE, "the line is too short to be correct: %.*s", len, line);
            const std::string cname = field(line, len, 20, 2);
            if (!chain || cname != chain->name) {
                if (!model) {
                    models.push_back(PModel{"1", {}});
                    model = &models.back();
                }
                model->chains.push_back(PChain{cname, {}});
                chain = &model->chains.back();
                res = nullptr;
            }
            const int num = read_seqnum(line, len, 22);
            const char icode = len > 26 ? line[26] : ' ';
            const std::string rname = field(line, len, 17, 3);
            const std::string seg = len > 72 ? field(line, len, 72, 4) : std::string();
            if (!res || res->num != num || res->icode != icode || res->name != rname || res->segment != seg) {
                res = nullptr;
                for (PRes& r : chain->res)
                    if (r.num == num && r.icode == icode && r.name == rname && r.segment == seg) { res = &r; break; }
                if (!res) {
                    chain->res.push_back(PRes{num, icode, rname, seg, 'A', {}});
                    res = &chain->res.back();
                }
                res->het = (char)(std::toupper((unsigned char)line[0]) == 'H' ? 'H' : 'A');
            }
            PAtom a;
            a.name = field(line, len, 12, 4);
            a.altloc = line[16] == ' ' ? '\0' : line[16];
            a.x = (float)read_double(line, len, 30, 8);
            a.y = (float)read_double(line, len, 38, 8);
            a.z = (float)read_double(line, len, 46, 8);
            char pad[4] = {' ', ' ', ' ', ' '};
            for (int i = 0; i < 4 && 12 + i < len; ++i) pad[i] = line[12 + i];
            if (len > 76 && (std::isalpha((unsigned char)line[76]) || (len > 77 && std::isalpha((unsigned char)line[77])))) {
                const char f[2] = {line[76], len > 77 ? line[77] : ' '};
                a.element = element_from_field(f);
            } else {
                a.element = element_from_padded_name(pad);
            }
            res->atoms.push_back(a);
        } else if (record_is(line, len, "MODE")) {
            if (model && chain) return fail(PESTO_IO_ERR_PARSE, "MODEL without ENDMDL?");
            const std::string name = std::to_string(read_int(line, len, 6, 8));
            model = nullptr;
            for (PModel& m : models) if (m.name == name) model = &m;
            if (model && !model->chains.empty()) return fail(PESTO_IO_ERR_PARSE, "duplicate MODEL number: %s", name.c_str());
            if (!model) { models.push_back(PModel{name, {}}); model = &models.back(); }
            chain = nullptr; res = nullptr;
        } else if (record_is(line, len, "ENDM")) {
            model = nullptr; chain = nullptr; res = nullptr;
        } else if (record_is(line, len, "END ")) {
            break;
        }
    }
    // merge chain parts of equal name, first appearance keeps its place
    for (PModel& m : models) {
        std::vector<PChain> merged;
        for (PChain& c : m.chains) {
            PChain* dst = nullptr;
            for (PChain& d : merged) if (d.name == c.name) { dst = &d; break; }
            if (!dst) merged.push_back(std::move(c));
            else for (PRes& r : c.res) dst->res.push_back(std::move(r));
        }
        m.chains.swap(merged);
    }
    return 0;
}

// read_pdb's own loop: src/structure_io.py:22-44
int models_to_atoms(const std::vector<PModel>& models, Atoms& out) {
    std::unordered_set<std::string> altloc_seen;   // keys carry no model / insertion code, exactly like the reference's list
    for (size_t mid = 0; mid < models.size(); ++mid)
        for (const PChain& c : models[mid].chains)
            for (const PRes& r : c.res)
                for (const PAtom& a : r.atoms) {
                    if (a.altloc) {
                        const std::string key = c.name + "_" + std::to_string(r.num) + "_" + a.name;
                        if (!altloc_seen.insert(key).second) continue;   // keep the first encountered
                    }
                    out.icode.push_back(r.icode == ' ' ? std::string() : std::string(1, r.icode));
                    out.element.push_back(a.element);
                    out.name.push_back(a.name);
                    out.xyz.push_back(a.x); out.xyz.push_back(a.y); out.xyz.push_back(a.z);
                    out.resname.push_back(r.name);
                    out.resid.push_back(r.num);
                    out.het.push_back(r.het);
                    out.chain.push_back(c.name + ":" + std::to_string(mid));
                }
    return 0;
}

// ------------------------------------------------------------------------------------------------ preprocessing
// clean_structure(structure, rm_wat=True): src/structure.py:14-56
int clean_structure(Atoms& a) {
    std::vector<size_t> keep;
    for (size_t i = 0; i < a.size(); ++i)
        if (a.resname[i] != "HOH" && a.element[i] != "H" && a.element[i] != "D" && a.resname[i] != "DOD") keep.push_back(i);
    if (keep.empty()) return fail(PESTO_IO_ERR_INVALID, "no atoms left after removing water and hydrogens");
    Atoms b = a.select(keep);
    // a new residue starts wherever the chain, the insertion code or the residue number changes
    int64_t r = 1;
    std::vector<int64_t> resid(b.size());
    for (size_t i = 0; i < b.size(); ++i) {
        if (i > 0 && (b.chain[i] != b.chain[i - 1] || b.resid[i] != b.resid[i - 1] || (b.has_icode && b.icode[i] != b.icode[i - 1]))) ++r;
        resid[i] = r;
    }
    b.resid.swap(resid);
    b.icode.clear();
    b.has_icode = false;
    a = std::move(b);
    return 0;
}

// tag_hetatm_chains: src/structure.py:95-110. The counter runs over ALL HETATM atoms of the structure and advances when
// the (renumbered) resid changes between consecutive HETATM atoms; numpy's '<U10' cast cuts every chain name to 10 chars.
void tag_hetatm_chains(Atoms& a) {
    int64_t counter = 0, prev = 0;
    bool first = true;
    for (size_t i = 0; i < a.size(); ++i) {
        if (a.het[i] == 'H') {
            if (!first && a.resid[i] != prev) ++counter;
            first = false;
            prev = a.resid[i];
            a.chain[i] += ":" + std::to_string(counter);
        }
        if (a.chain[i].size() > 10) a.chain[i].resize(10);
    }
}

// split_by_chain (src/structure.py:63-80) keeps a dict keyed by np.unique(chain_name), i.e. in sorted-name order
std::vector<std::pair<std::string, std::vector<size_t>>> split_by_chain(const Atoms& a) {
    std::map<std::string, std::vector<size_t>> m;
    for (size_t i = 0; i < a.size(); ++i) m[a.chain[i]].push_back(i);
    return {m.begin(), m.end()};
}

int preprocess(Atoms& a, int steps) {
    if (steps & PESTO_IO_CLEAN)
        if (int rc = clean_structure(a)) return rc;
    if (steps & PESTO_IO_TAG_HETATM) tag_hetatm_chains(a);
    if (!(steps & (PESTO_IO_SPLIT | PESTO_IO_FILTER_NON_ATOMIC | PESTO_IO_REMOVE_DUPLICATES))) return 0;
    auto subunits = split_by_chain(a);
    if (steps & PESTO_IO_FILTER_NON_ATOMIC) {   // src/structure.py:137-146
        std::vector<std::pair<std::string, std::vector<size_t>>> kept;
        for (auto& su : subunits) {
            std::set<int64_t> res;
            for (size_t i : su.second) res.insert(a.resid[i]);
            if (!(su.second.size() == res.size() && su.second.size() > 1)) kept.push_back(std::move(su));
        }
        subunits.swap(kept);
    }
    if (steps & PESTO_IO_REMOVE_DUPLICATES) {   // src/structure.py:113-134
        std::vector<size_t> tagged;
        for (size_t s = 0; s < subunits.size(); ++s)
            if (std::count(subunits[s].first.begin(), subunits[s].first.end(), ':') == 2) tagged.push_back(s);
        std::vector<bool> dead(subunits.size(), false);
        for (size_t x = 0; x < tagged.size(); ++x)
            for (size_t y = x + 1; y < tagged.size(); ++y) {
                const size_t si = tagged[x], sj = tagged[y];
                if (dead[si] || dead[sj]) continue;
                const auto &I = subunits[si].second, &J = subunits[sj].second;
                if (I.size() != J.size()) continue;
                float dmin = INFINITY;   // float32 arithmetic like numpy on float32 coordinates
                for (size_t t = 0; t < I.size(); ++t) {
                    const float dx = a.xyz[3 * I[t]] - a.xyz[3 * J[t]], dy = a.xyz[3 * I[t] + 1] - a.xyz[3 * J[t] + 1],
                                dz = a.xyz[3 * I[t] + 2] - a.xyz[3 * J[t] + 2];
                    const float xx = dx * dx, yy = dy * dy, zz = dz * dz;
                    const float d = std::sqrt((xx + yy) + zz);
                    dmin = std::min(dmin, d);
                }
                if ((double)dmin < 0.2) dead[sj] = true;
            }
        std::vector<std::pair<std::string, std::vector<size_t>>> kept;
        for (size_t s = 0; s < subunits.size(); ++s)
            if (!dead[s]) kept.push_back(std::move(subunits[s]));
        subunits.swap(kept);
    }
    if (subunits.empty()) return fail(PESTO_IO_ERR_INVALID, "no subunits left after filtering");
    // concatenate_chains (src/structure.py:83-92): subunits in dict (= sorted name) order
    std::vector<size_t> order;
    for (auto& su : subunits) order.insert(order.end(), su.second.begin(), su.second.end());
    a = a.select(order);
    return 0;
}

// ------------------------------------------------------------------------------------------------ encoding
// std_elements / std_resnames / std_names: src/data_encoding.py:6-31 (the one-hot vocabularies of the trained models)
const char* const STD_ELEMENTS[] = {"C",  "O",  "N",  "S",  "P",  "Se", "Mg", "Cl", "Zn", "Fe", "Ca", "Na", "F",  "Mn", "I",
                                    "K",  "Br", "Cu", "Cd", "Ni", "Co", "Sr", "Hg", "W",  "As", "B",  "Mo", "Ba", "Pt"};
const char* const STD_RESNAMES[] = {"LEU", "GLU", "ARG", "LYS", "VAL", "ILE", "PHE", "ASP", "TYR", "ALA", "THR", "SER", "GLN", "ASN",
                                    "PRO", "GLY", "HIS", "TRP", "MET", "CYS", "G",   "A",   "C",   "U",   "DG",  "DA",  "DT",  "DC"};
const char* const STD_NAMES[] = {"CA",  "N",   "C",   "O",   "CB",  "CG",  "CD2", "CD1", "CG1", "CG2", "CD",  "OE1", "OE2", "OG",  "OG1", "OD1",
                                 "OD2", "CE",  "NZ",  "NE",  "CZ",  "NH2", "NH1", "ND2", "CE2", "CE1", "NE2", "OH",  "ND1", "SD",  "SG",  "NE1",
                                 "CE3", "CZ3", "CZ2", "CH2", "P",   "C3'", "C4'", "O3'", "C5'", "O5'", "O4'", "C1'", "C2'", "O2'", "OP1", "OP2",
                                 "N9",  "N2",  "O6",  "N7",  "C8",  "N1",  "N3",  "C2",  "C4",  "C6",  "C5",  "N6",  "N4",  "O2",  "O4"};
template <size_t K>
int vocab_index(const char* const (&v)[K], const std::string& s) {   // onehot(): index, or K = the "unknown" column
    for (size_t i = 0; i < K; ++i)
        if (s == v[i]) return (int)i;
    return (int)K;
}

// rank of each atom's resid among the sorted unique resids = its column in M (src/data_encoding.py:72-73)
int64_t residue_columns(const Atoms& a, std::vector<int32_t>& col) {
    std::vector<int64_t> u(a.resid);
    std::sort(u.begin(), u.end());
    u.erase(std::unique(u.begin(), u.end()), u.end());
    col.resize(a.size());
    for (size_t i = 0; i < a.size(); ++i) col[i] = (int32_t)(std::lower_bound(u.begin(), u.end(), a.resid[i]) - u.begin());
    return (int64_t)u.size();
}

// ------------------------------------------------------------------------------------------------ writing
// ---- number formatting without printf (the writer is the slowest step of the native apply_model chain otherwise)
// "{:>Wd}"
void append_int(std::string& out, long long v, int width) {
    char buf[24];
    int n = 0;
    const bool neg = v < 0;
    unsigned long long u = neg ? 0ull - (unsigned long long)v : (unsigned long long)v;
    do { buf[n++] = (char)('0' + u % 10); u /= 10; } while (u);
    if (neg) buf[n++] = '-';
    for (int i = n; i < width; ++i) out.push_back(' ');
    while (n) out.push_back(buf[--n]);
}
// "{:W.Df}" of a value that came from a float32 (D <= 3): v * 10^D has at most 24 + 10 significant bits, so the product is exact
// in double and rint (ties to even) reproduces printf's correctly rounded decimal, including exact ties. Anything else
// (non-finite, huge) goes through snprintf.
void append_fixed(std::string& out, double v, int width, int decimals) {
    static const double P10[4] = {1.0, 10.0, 100.0, 1000.0};
    if (!(std::fabs(v) < 1e12) || decimals > 3) {
        char buf[64];
        snprintf(buf, sizeof buf, "%*.*f", width, decimals, v);
        out += buf;
        return;
    }
    const bool neg = std::signbit(v);
    unsigned long long n = (unsigned long long)std::llrint(std::fabs(v) * P10[decimals]);
    char buf[32];
    int k = 0;
    for (int d = 0; d < decimals; ++d) { buf[k++] = (char)('0' + n % 10); n /= 10; }
    if (decimals) buf[k++] = '.';
    do { buf[k++] = (char)('0' + n % 10); n /= 10; } while (n);
    if (neg) buf[k++] = '-';
    for (int i = k; i < width; ++i) out.push_back(' ');
    while (k) out.push_back(buf[--k]);
}
void append_left(std::string& out, const std::string& t, size_t width) {      // "{:<Ws}"
    out += t;
    for (size_t i = t.size(); i < width; ++i) out.push_back(' ');
}
void append_right(std::string& out, const std::string& t, size_t width) {     // "{:>Ws}"
    for (size_t i = t.size(); i < width; ++i) out.push_back(' ');
    out += t;
}

// save_pdb(split_by_chain(structure), path): src/structure_io.py:96-123
// line = "{:<6s}{:>5d} {:<4s} {:>3s} {:1s}{:>4d}    {:8.3f}{:8.3f}{:8.3f}{:6.2f}{:6.2f}          {:<2s}  "
void format_pdb(const Atoms& a, const std::vector<float>& bf, std::string& out) {
    out.clear();
    out.reserve(a.size() * 82 + 64);
    for (const auto& su : split_by_chain(a)) {
        const std::string head = su.first.substr(0, su.first.find(':'));
        const char c = head.empty() ? ' ' : head[0];
        long long serial = 0;
        for (size_t i : su.second) {
            const double b = bf.empty() ? 0.0 : (double)bf[i];
            out += a.het[i] == 'A' ? "ATOM  " : "HETATM";
            append_int(out, ++serial, 5);
            out.push_back(' ');
            append_left(out, a.name[i], 4);
            out.push_back(' ');
            append_right(out, a.resname[i], 3);
            out.push_back(' ');
            out.push_back(c);
            append_int(out, a.resid[i], 4);
            out += "    ";
            append_fixed(out, (double)a.xyz[3 * i], 8, 3);
            append_fixed(out, (double)a.xyz[3 * i + 1], 8, 3);
            append_fixed(out, (double)a.xyz[3 * i + 2], 8, 3);
            append_fixed(out, b, 6, 2);
            append_fixed(out, b, 6, 2);
            out += "          ";
            append_left(out, a.element[i], 2);
            out += "  \n";
        }
        out += "TER\n";
    }
    out += "END";
}

// encode_bfactor: per-atom values as they are; per-residue values expanded through the sorted unique resids
int expand_bfactor(const Atoms& a, const float* bf, int64_t n_values, std::vector<float>& out) {
    out.clear();
    if (!bf) return 0;
    if (n_values == (int64_t)a.size()) { out.assign(bf, bf + n_values); return 0; }
    std::vector<int32_t> col;
    const int64_t n_res = residue_columns(a, col);
    if (n_values != n_res)
        return fail(PESTO_IO_ERR_INVALID, "bfactor has %lld values; the structure has %lld atoms and %lld residues", (long long)n_values,
                    (long long)a.size(), (long long)n_res);
    out.resize(a.size());
    for (size_t i = 0; i < a.size(); ++i) out[i] = bf[col[i]];
    return 0;
}

}  // namespace

struct pesto_structure {
    Atoms a;
    std::string text;   // last pesto_io_format_pdb result
};

extern "C" {

const char* pesto_io_last_error(void) { return g_err.c_str(); }

int pesto_io_parse_pdb(const char* text, int64_t len, pesto_structure** out) {
    if (!text || len < 0 || !out) return fail(PESTO_IO_ERR_INVALID, "bad arguments");
    *out = nullptr;
    std::vector<PModel> models;
    if (int rc = parse_pdb(text, len, models)) return rc;
    pesto_structure* s = new pesto_structure();
    models_to_atoms(models, s->a);
    *out = s;
    return 0;
}

int pesto_io_read_pdb(const char* path, pesto_structure** out) {
    if (!path || !out) return fail(PESTO_IO_ERR_INVALID, "bad arguments");
    *out = nullptr;
    FILE* f = fopen(path, "rb");
    if (!f) return fail(PESTO_IO_ERR_FILE, "cannot open %s", path);
    std::string buf;
    char chunk[1 << 16];
    size_t n;
    while ((n = fread(chunk, 1, sizeof chunk, f)) > 0) buf.append(chunk, n);
    fclose(f);
    return pesto_io_parse_pdb(buf.data(), (int64_t)buf.size(), out);
}

int pesto_io_from_arrays(int64_t n, const float* xyz, const int64_t* resid, const char* const text[6], const int32_t width[6],
                         pesto_structure** out) {
    if (n < 0 || !xyz || !resid || !text || !width || !out) return fail(PESTO_IO_ERR_INVALID, "bad arguments");
    *out = nullptr;
    for (int f = 0; f < 5; ++f)
        if (!text[f] || width[f] < 1) return fail(PESTO_IO_ERR_INVALID, "text field %d missing", f);
    pesto_structure* s = new pesto_structure();
    Atoms& a = s->a;
    a.xyz.assign(xyz, xyz + 3 * n);
    a.resid.assign(resid, resid + n);
    auto get = [&](int f, int64_t i) {
        const char* p = text[f] + (size_t)i * width[f];
        return std::string(p, strnlen(p, width[f]));
    };
    a.has_icode = text[PESTO_IO_ICODE] != nullptr;
    for (int64_t i = 0; i < n; ++i) {
        a.name.push_back(get(PESTO_IO_NAME, i));
        a.element.push_back(get(PESTO_IO_ELEMENT, i));
        a.resname.push_back(get(PESTO_IO_RESNAME, i));
        a.het.push_back(get(PESTO_IO_HET_FLAG, i) == "H" ? 'H' : 'A');
        a.chain.push_back(get(PESTO_IO_CHAIN_NAME, i));
        if (a.has_icode) a.icode.push_back(get(PESTO_IO_ICODE, i));
    }
    *out = s;
    return 0;
}

int pesto_io_free(pesto_structure* s) {
    delete s;
    return 0;
}

int pesto_io_preprocess(pesto_structure* s, int32_t steps) {
    if (!s) return fail(PESTO_IO_ERR_INVALID, "null structure");
    if (steps & ~PESTO_IO_ALL) return fail(PESTO_IO_ERR_INVALID, "unknown preprocessing step bits %d", steps);
    return preprocess(s->a, steps);
}

int pesto_io_n_atoms(const pesto_structure* s, int64_t* n) {
    if (!s || !n) return fail(PESTO_IO_ERR_INVALID, "bad arguments");
    *n = (int64_t)s->a.size();
    return 0;
}

int pesto_io_get_xyz(const pesto_structure* s, float* xyz) {
    if (!s || !xyz) return fail(PESTO_IO_ERR_INVALID, "bad arguments");
    std::copy(s->a.xyz.begin(), s->a.xyz.end(), xyz);
    return 0;
}

int pesto_io_get_resid(const pesto_structure* s, int64_t* resid) {
    if (!s || !resid) return fail(PESTO_IO_ERR_INVALID, "bad arguments");
    std::copy(s->a.resid.begin(), s->a.resid.end(), resid);
    return 0;
}

int pesto_io_get_text(const pesto_structure* s, int32_t field, char* out, int32_t width) {
    if (!s || !out || width < 1) return fail(PESTO_IO_ERR_INVALID, "bad arguments");
    const Atoms& a = s->a;
    const std::vector<std::string>* v = nullptr;
    switch (field) {
        case PESTO_IO_NAME: v = &a.name; break;
        case PESTO_IO_ELEMENT: v = &a.element; break;
        case PESTO_IO_RESNAME: v = &a.resname; break;
        case PESTO_IO_CHAIN_NAME: v = &a.chain; break;
        case PESTO_IO_ICODE:
            if (!a.has_icode) return fail(PESTO_IO_ERR_INVALID, "the structure has no icode field any more (clean_structure drops it)");
            v = &a.icode;
            break;
        case PESTO_IO_HET_FLAG: break;
        default: return fail(PESTO_IO_ERR_INVALID, "unknown text field %d", field);
    }
    memset(out, 0, (size_t)width * a.size());
    for (size_t i = 0; i < a.size(); ++i) {
        if (!v) { out[(size_t)i * width] = a.het[i]; continue; }
        const std::string& t = (*v)[i];
        if ((int)t.size() > width) return fail(PESTO_IO_ERR_INVALID, "value '%s' does not fit in %d bytes", t.c_str(), width);
        memcpy(out + (size_t)i * width, t.data(), t.size());
    }
    return 0;
}

int pesto_io_encode(const pesto_structure* s, int32_t n0, float* X, float* q0, int32_t* res_of_atom, int64_t* n_res) {
    if (!s) return fail(PESTO_IO_ERR_INVALID, "null structure");
    if (n0 != 30 && n0 != 123) return fail(PESTO_IO_ERR_INVALID, "n0 must be 30 (elements) or 123 (elements | resnames | names)");
    const Atoms& a = s->a;
    if (X) std::copy(a.xyz.begin(), a.xyz.end(), X);
    if (q0) {
        std::fill(q0, q0 + a.size() * (size_t)n0, 0.0f);
        for (size_t i = 0; i < a.size(); ++i) {
            float* row = q0 + i * (size_t)n0;
            row[vocab_index(STD_ELEMENTS, a.element[i])] = 1.0f;
            if (n0 == 123) {
                row[30 + vocab_index(STD_RESNAMES, a.resname[i])] = 1.0f;
                row[59 + vocab_index(STD_NAMES, a.name[i])] = 1.0f;
            }
        }
    }
    if (res_of_atom || n_res) {
        std::vector<int32_t> col;
        const int64_t r = residue_columns(a, col);
        if (res_of_atom) std::copy(col.begin(), col.end(), res_of_atom);
        if (n_res) *n_res = r;
    }
    return 0;
}

int pesto_io_format_pdb(pesto_structure* s, const float* bfactor, int64_t n_values, const char** text, int64_t* len) {
    if (!s || !text || !len) return fail(PESTO_IO_ERR_INVALID, "bad arguments");
    std::vector<float> bf;
    if (int rc = expand_bfactor(s->a, bfactor, n_values, bf)) return rc;
    format_pdb(s->a, bf, s->text);
    *text = s->text.c_str();
    *len = (int64_t)s->text.size();
    return 0;
}

int pesto_io_write_pdb(const pesto_structure* s, const float* bfactor, int64_t n_values, const char* path) {
    if (!s || !path) return fail(PESTO_IO_ERR_INVALID, "bad arguments");
    std::vector<float> bf;
    if (int rc = expand_bfactor(s->a, bfactor, n_values, bf)) return rc;
    std::string text;
    format_pdb(s->a, bf, text);
    FILE* f = fopen(path, "wb");
    if (!f) return fail(PESTO_IO_ERR_FILE, "cannot open %s for writing", path);
    const bool ok = fwrite(text.data(), 1, text.size(), f) == text.size();
    if (fclose(f) != 0 || !ok) return fail(PESTO_IO_ERR_FILE, "short write to %s", path);
    return 0;
}


/* dense residue mask -> residue column per atom, one pass over the rows (the numpy form makes four) */
int pesto_io_mask_to_segments(const float* M, int64_t N, int64_t R, int32_t* res_of_atom) {
    return pesto_io_mask_to_segments_any(M, 4, N, R, res_of_atom);
}

int pesto_io_mask_to_segments_any(const void* M, int32_t elem_bytes, int64_t N, int64_t R, int32_t* res_of_atom) {
    if (!M || !res_of_atom || N < 1 || R < 1 || R > 0x7fffffff) return fail(PESTO_IO_ERR_INVALID, "bad arguments");
    if (elem_bytes != 1 && elem_bytes != 4) return fail(PESTO_IO_ERR_INVALID, "mask elements must be 1 byte (bool / uint8) or 4 bytes (float32)");
    std::vector<unsigned char> seen((size_t)R, 0);
    const bool wide = have_avx2();
    int64_t hint = 0;      // column of the previous row: the reference's columns are unique(resid) of a contiguously numbered structure, so a
                           // row's member is the previous row's column or the next one (src/structure.py:47, src/data_encoding.py:73) - tried first
    for (int64_t i = 0; i < N; ++i) {
        int64_t at = -1;
        if (elem_bytes == 1) {
            const uint8_t* row = (const uint8_t*)M + i * R;
            // EXACT test of the whole row in one pass of byte sums (no per-element branch): the bytes of a valid row add up to 1
            const uint64_t total = wide ? row_sum_u8_avx2(row, R) : row_sum_u8(row, R);
            if (total != 1) {
                int members = 0;
                for (int64_t r = 0; r < R; ++r) members += row[r] != 0;
                if (members != 1) return fail(PESTO_IO_ERR_INVALID, "M: atom %lld belongs to %d residues (every atom must belong to exactly one)", (long long)i, members);
            }
            if (row[hint] != 0) at = hint;
            else if (hint + 1 < R && row[hint + 1] != 0) at = hint + 1;
            else for (int64_t r = 0; r < R; ++r) if (row[r] != 0) { at = r; break; }
        } else {
            const float* row = (const float*)M + i * R;
            // a valid row is one 1.0f among +0.0f: as integers its words add up to exactly 0x3f800000 and no word but the member's is
            // set. Anything else (other values above 0.5, -0.0f, several members) takes the element-wise statement of the contract.
            const uint64_t total = wide ? row_sum_u32_avx2((const uint32_t*)row, R) : row_sum_u32((const uint32_t*)row, R);
            if (total == 0x3f800000u) {
                if (row[hint] == 1.0f) at = hint;
                else if (hint + 1 < R && row[hint + 1] == 1.0f) at = hint + 1;
                else for (int64_t r = 0; r < R; ++r) if (row[r] == 1.0f) { at = r; break; }
            }
            if (at < 0) {      // (also reached by a row whose words happen to add up to 0x3f800000 without holding a 1.0f)
                int members = 0;
                for (int64_t r = 0; r < R; ++r) if (row[r] > 0.5f) { ++members; at = r; }
                if (members != 1) return fail(PESTO_IO_ERR_INVALID, "M: atom %lld belongs to %d residues (every atom must belong to exactly one)", (long long)i, members);
            }
        }
        res_of_atom[i] = (int32_t)at;
        seen[(size_t)at] = 1;
        hint = at;
    }
    for (int64_t r = 0; r < R; ++r)
        if (!seen[(size_t)r]) return fail(PESTO_IO_ERR_INVALID, "M: residue column %lld is empty", (long long)r);
    return 0;
}

}  // extern "C"
