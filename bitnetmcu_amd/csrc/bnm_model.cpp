// Run-time loader for exporter-written BitNetMCU_model.h text, and the BNMBLOB1 binary form.
// Format reference: exportquant.py:49-263 (writer); dialects: BitNetMCU_model_fc.h:8-25
// (8 words per line, MODEL_ define before the include guard), mcu/BitNetMCU_model_12k.h:8-31
// (single-line arrays, un-padded hex such as 0x8aa99ba, embedded /* */ statistics comment),
// BitNetMCU_model_cnn.h:12-34 (conv/pool geometry macros, int8_t arrays in decimal),
// uint16_t ternary arrays (exportquant.py:161-174).
#include "bnm_model.hpp"
#include <cctype>
#include <cstdlib>
#include <cstring>
#include <map>

int bnm_codec_field_bits(int32_t bpw) {
    switch (bpw) {
        case 1: return 1;
        case 2: return 2;
        case 4: case 12: case 20: return 4;
        case 16: return 8;
        default: return 0;
    }
}

bool bnm_codec_known(int32_t bpw) { return bpw == 64 || bnm_codec_field_bits(bpw) != 0; }

uint64_t bnm_fc_weight_count(int32_t bpw, uint32_t n_in, uint32_t n_out) {
    if (bpw == 64) return (uint64_t)n_out * (n_in / 10u);
    int fb = bnm_codec_field_bits(bpw);
    if (!fb) return 0;
    uint32_t per_word = 32u / (uint32_t)fb;
    return (uint64_t)n_out * ((n_in + per_word - 1u) / per_word);
}

uint32_t bnm_fc_real_inputs(const bnm_layer_info &li, uint32_t prev_outputs) {
    if (li.bits_per_weight == 64 && li.n_input >= prev_outputs && li.n_input - prev_outputs < 10u)
        return prev_outputs;
    return li.n_input;
}

std::vector<uint32_t> bnm_model::fc_layers() const {
    std::vector<uint32_t> v;
    for (uint32_t i = 0; i < layers.size(); i++)
        if (layers[i].info.type == BNM_LAYER_FC) v.push_back(i);
    return v;
}

uint32_t bnm_model::num_classes() const {
    auto f = fc_layers();
    return f.empty() ? 0 : layers[f.back()].info.n_output;
}

namespace {

struct ArrayDecl {
    std::string ctype;
    std::vector<int64_t> values;
};

std::string strip_comments(const char *t, size_t n) {
    std::string o;
    o.reserve(n);
    size_t i = 0;
    while (i < n) {
        if (t[i] == '/' && i + 1 < n && t[i + 1] == '/') {
            while (i < n && t[i] != '\n') i++;
        } else if (t[i] == '/' && i + 1 < n && t[i + 1] == '*') {
            i += 2;
            while (i + 1 < n && !(t[i] == '*' && t[i + 1] == '/')) i++;
            i = (i + 2 <= n) ? i + 2 : n;
            o.push_back(' ');
        } else {
            o.push_back(t[i++]);
        }
    }
    return o;
}

bool is_ident(char c) { return std::isalnum((unsigned char)c) || c == '_'; }

// "L<digits>" ?
bool layer_prefix(const std::string &s, uint32_t &order) {
    if (s.size() < 2 || s[0] != 'L') return false;
    for (size_t i = 1; i < s.size(); i++)
        if (!std::isdigit((unsigned char)s[i])) return false;
    order = (uint32_t)std::strtoul(s.c_str() + 1, nullptr, 10);
    return true;
}

}  // namespace

bool bnm_parse_header_text(const char *text, size_t len, bnm_model &out, std::string &err) {
    std::string src = strip_comments(text, len);
    std::map<std::string, std::string> defs;
    std::vector<std::string> def_order;
    std::map<std::string, ArrayDecl> arrays;

    size_t i = 0, n = src.size();
    while (i < n) {
        // skip whitespace
        while (i < n && std::isspace((unsigned char)src[i])) i++;
        if (i >= n) break;
        if (src[i] == '#') {
            size_t e = src.find('\n', i);
            if (e == std::string::npos) e = n;
            std::string line = src.substr(i, e - i);
            i = e;
            size_t p = 1;
            while (p < line.size() && std::isspace((unsigned char)line[p])) p++;
            if (line.compare(p, 6, "define") != 0) continue;
            p += 6;
            while (p < line.size() && std::isspace((unsigned char)line[p])) p++;
            size_t q = p;
            while (q < line.size() && is_ident(line[q])) q++;
            std::string name = line.substr(p, q - p);
            while (q < line.size() && std::isspace((unsigned char)line[q])) q++;
            std::string val = line.substr(q);
            while (!val.empty() && std::isspace((unsigned char)val.back())) val.pop_back();
            if (!name.empty()) {
                if (!defs.count(name)) def_order.push_back(name);
                defs[name] = val;
            }
            continue;
        }
        // a declaration: read up to ';'
        size_t e = src.find(';', i);
        if (e == std::string::npos) break;
        std::string decl = src.substr(i, e - i);
        i = e + 1;
        size_t br = decl.find('{');
        size_t sq = decl.find('[');
        if (br == std::string::npos || sq == std::string::npos || sq > br) continue;
        // name = identifier right before '['
        size_t q = sq;
        while (q > 0 && std::isspace((unsigned char)decl[q - 1])) q--;
        size_t p = q;
        while (p > 0 && is_ident(decl[p - 1])) p--;
        std::string name = decl.substr(p, q - p);
        ArrayDecl a;
        if (decl.find("uint32_t") < p) a.ctype = "uint32_t";
        else if (decl.find("uint16_t") < p) a.ctype = "uint16_t";
        else if (decl.find("int8_t") < p) a.ctype = "int8_t";
        else continue;
        const char *c = decl.c_str() + br + 1;
        const char *end = decl.c_str() + decl.size();
        while (c < end) {
            while (c < end && (std::isspace((unsigned char)*c) || *c == ',' || *c == '}')) c++;
            if (c >= end) break;
            char *stop = nullptr;
            long long v = std::strtoll(c, &stop, 0);
            if (stop == c) {
                err = "array " + name + ": cannot parse element near '" + std::string(c, std::min<size_t>(12, end - c)) + "'";
                return false;
            }
            // tolerate "12.0"-style literals (mcu/BitNetMCUdemo.c:23 writes int8 data that way)
            if (stop < end && *stop == '.') {
                stop++;
                while (stop < end && std::isdigit((unsigned char)*stop)) stop++;
            }
            a.values.push_back(v);
            c = stop;
        }
        arrays[name] = std::move(a);
    }

    out = bnm_model();
    out.kind = defs.count("MODEL_CNNMNIST") ? BNM_KIND_CNN : BNM_KIND_FC;

    auto geti = [&](const std::string &k, uint32_t &v) -> bool {
        auto it = defs.find(k);
        if (it == defs.end() || it->second.empty()) return false;
        char *stop = nullptr;
        long long x = std::strtoll(it->second.c_str(), &stop, 0);
        if (stop == it->second.c_str()) return false;
        v = (uint32_t)x;
        return true;
    };

    for (const std::string &d : def_order) {
        if (d.size() < 8 || d.compare(d.size() - 7, 7, "_active") != 0) continue;
        std::string p = d.substr(0, d.size() - 7);
        uint32_t order = 0;
        if (!layer_prefix(p, order)) continue;
        BnmLayer L;
        L.info.order = order;
        std::string type = defs.count(p + "_type") ? defs[p + "_type"] : "";
        if (type == "MaxPool2d") {
            L.info.type = BNM_LAYER_POOL;
            if (!geti(p + "_pool_size", L.info.pool_size) || !geti(p + "_incoming_x", L.info.incoming_x) ||
                !geti(p + "_outgoing_x", L.info.outgoing_x)) {
                err = p + ": incomplete MaxPool2d macros";
                return false;
            }
        } else if (type == "BitConv2d") {
            L.info.type = BNM_LAYER_CONV;
            uint32_t bpw = 0;
            if (!geti(p + "_in_channels", L.info.in_channels) || !geti(p + "_out_channels", L.info.out_channels) ||
                !geti(p + "_incoming_x", L.info.incoming_x) || !geti(p + "_outgoing_x", L.info.outgoing_x) ||
                !geti(p + "_kernel_size", L.info.kernel_size) || !geti(p + "_groups", L.info.groups) ||
                !geti(p + "_bitperweight", bpw)) {
                err = p + ": incomplete BitConv2d macros";
                return false;
            }
            L.info.bits_per_weight = (int32_t)bpw;
            auto it = arrays.find(p + "_weights");
            if (it == arrays.end() || it->second.ctype != "int8_t") {
                err = p + ": missing int8_t weight array";
                return false;
            }
            L.info.weight_elem_bytes = 1;
            L.info.weight_count = (uint32_t)it->second.values.size();
            L.weights.resize(it->second.values.size());
            for (size_t k = 0; k < it->second.values.size(); k++) L.weights[k] = (uint8_t)(int8_t)it->second.values[k];
        } else if (type.empty()) {
            L.info.type = BNM_LAYER_FC;
            uint32_t bpw = 0;
            if (!geti(p + "_bitperweight", bpw) || !geti(p + "_incoming_weights", L.info.n_input) ||
                !geti(p + "_outgoing_weights", L.info.n_output)) {
                err = p + ": incomplete FC macros";
                return false;
            }
            L.info.bits_per_weight = (int32_t)bpw;
            auto it = arrays.find(p + "_weights");
            if (it == arrays.end()) {
                err = p + ": missing weight array";
                return false;
            }
            const ArrayDecl &a = it->second;
            uint32_t eb = a.ctype == "uint32_t" ? 4 : a.ctype == "uint16_t" ? 2 : 1;
            L.info.weight_elem_bytes = eb;
            L.info.weight_count = (uint32_t)a.values.size();
            L.weights.resize(a.values.size() * eb);
            for (size_t k = 0; k < a.values.size(); k++) {
                uint64_t v = (uint64_t)a.values[k];
                std::memcpy(&L.weights[k * eb], &v, eb);  // little-endian host
            }
        } else {
            err = p + ": unknown layer type '" + type + "'";
            return false;
        }
        out.layers.push_back(std::move(L));
    }
    if (out.layers.empty()) {
        err = "no Lk_active layer found in header text";
        return false;
    }
    return bnm_validate_schedule(out, err);
}

bool bnm_validate_schedule(const bnm_model &m, std::string &err) {
    size_t pos = 0;
    uint32_t width = 256;
    if (m.kind == BNM_KIND_CNN) {
        // BitNetMCU_MNIST_dll.c:66-77: conv(16) conv(14) pool(12) conv(6) pool(4), per channel
        static const uint32_t kinds[5] = {BNM_LAYER_CONV, BNM_LAYER_CONV, BNM_LAYER_POOL, BNM_LAYER_CONV, BNM_LAYER_POOL};
        static const uint32_t inx[5] = {16, 14, 12, 6, 4};
        if (m.layers.size() < 6) { err = "CNN model: expected 5 conv/pool layers + FC layers"; return false; }
        uint32_t C = m.layers[0].info.out_channels;
        for (int k = 0; k < 5; k++) {
            const bnm_layer_info &li = m.layers[k].info;
            if (li.type != kinds[k] || li.incoming_x != inx[k]) {
                err = "CNN model: layer " + std::to_string(k) + " does not match the reference wrapper's conv/pool schedule";
                return false;
            }
            if (li.type == BNM_LAYER_CONV) {
                bool first = (k == 0);
                if (li.kernel_size != 3 || li.out_channels != C || li.in_channels != (first ? 1u : C) ||
                    li.groups != (first ? 1u : C) || li.weight_count != 9u * C || li.weight_elem_bytes != 1u ||
                    m.layers[k].weights.size() < 9u * (size_t)C) {
                    err = "CNN model: conv layer " + std::to_string(k) + " is not a 3x3 depthwise stage";
                    return false;
                }
            } else if (li.pool_size != 2) {
                err = "CNN model: only 2x2 pooling is supported";
                return false;
            }
        }
        if (C == 0 || C * 4u > 1024u) { err = "CNN model: unsupported channel count"; return false; }
        width = C * 4u;
        pos = 5;
    }
    if (pos >= m.layers.size()) { err = "model has no FC layer"; return false; }
    for (; pos < m.layers.size(); pos++) {
        const bnm_layer_info &li = m.layers[pos].info;
        if (li.type != BNM_LAYER_FC) { err = "conv/pool layer after the FC stack is not supported"; return false; }
        uint32_t real = bnm_fc_real_inputs(li, width);
        if (real != width) {
            err = "FC layer L" + std::to_string(li.order) + ": incoming_weights " + std::to_string(li.n_input) +
                  " does not match previous width " + std::to_string(width);
            return false;
        }
        // (real inputs: a ternary layer declares its input count padded to a multiple of 10 - 1024 real inputs read 1030)
        if (li.n_output == 0 || li.n_output > 1024 || real > 1024) { err = "FC layer too wide (max 1024)"; return false; }
        if (bnm_codec_known(li.bits_per_weight)) {
            int fb = bnm_codec_field_bits(li.bits_per_weight);
            if (fb && (li.n_input * (uint32_t)fb) % 32u != 0) {
                err = "FC layer L" + std::to_string(li.order) + ": incoming_weights*bits not a multiple of 32 (exportquant.py:97-98)";
                return false;
            }
            if (li.bits_per_weight == 64 && li.n_input % 10u != 0) { err = "ternary layer: incoming_weights must be a multiple of 10"; return false; }
            uint64_t need = bnm_fc_weight_count(li.bits_per_weight, li.n_input, li.n_output);
            // the kernels read `need` elements of the codec's own width (uint16 for ternary, uint32 otherwise,
            // exportquant.py:161-174,193-207): the declared C type must be that width and the bytes must be there
            const uint32_t want_eb = li.bits_per_weight == 64 ? 2u : 4u;
            if (li.weight_elem_bytes != want_eb) {
                err = "FC layer L" + std::to_string(li.order) + ": weight array declared with " + std::to_string(li.weight_elem_bytes) +
                      "-byte elements, the codec needs " + std::to_string(want_eb);
                return false;
            }
            if (li.weight_count < need || m.layers[pos].weights.size() < need * want_eb) {
                err = "FC layer L" + std::to_string(li.order) + ": weight array too short";
                return false;
            }
        }
        width = li.n_output;
    }
    return true;
}

// ------------------------------------------------------------------------------------------
// BNMBLOB1: [magic 8][version u32][kind u32][n_layers u32][total u32] then per layer
// {bnm_layer_info, offset u32, bytes u32}, then 16-byte aligned payloads.
// ------------------------------------------------------------------------------------------
namespace {
struct BlobHead {
    char magic[8];
    uint32_t version, kind, n_layers, total;
};
struct BlobLayer {
    bnm_layer_info info;
    uint32_t offset, bytes;
};
const char kMagic[8] = {'B', 'N', 'M', 'B', 'L', 'O', 'B', '1'};
}  // namespace

std::vector<uint8_t> bnm_serialize(const bnm_model &m) {
    size_t off = sizeof(BlobHead) + m.layers.size() * sizeof(BlobLayer);
    off = (off + 15) & ~size_t(15);
    std::vector<BlobLayer> bl(m.layers.size());
    for (size_t i = 0; i < m.layers.size(); i++) {
        bl[i].info = m.layers[i].info;
        bl[i].offset = (uint32_t)off;
        bl[i].bytes = (uint32_t)m.layers[i].weights.size();
        off = (off + bl[i].bytes + 15) & ~size_t(15);
    }
    std::vector<uint8_t> out(off, 0);
    BlobHead h;
    std::memcpy(h.magic, kMagic, 8);
    h.version = 1;
    h.kind = m.kind;
    h.n_layers = (uint32_t)m.layers.size();
    h.total = (uint32_t)off;
    std::memcpy(out.data(), &h, sizeof h);
    std::memcpy(out.data() + sizeof h, bl.data(), bl.size() * sizeof(BlobLayer));
    for (size_t i = 0; i < m.layers.size(); i++)
        if (bl[i].bytes) std::memcpy(out.data() + bl[i].offset, m.layers[i].weights.data(), bl[i].bytes);
    return out;
}

bool bnm_deserialize(const void *blob, size_t len, bnm_model &out, std::string &err) {
    if (len < sizeof(BlobHead)) { err = "blob too short"; return false; }
    BlobHead h;
    std::memcpy(&h, blob, sizeof h);
    if (std::memcmp(h.magic, kMagic, 8) != 0 || h.version != 1) { err = "not a BNMBLOB1 blob"; return false; }
    if (h.total > len || h.n_layers > 64 || sizeof(BlobHead) + (size_t)h.n_layers * sizeof(BlobLayer) > len) {
        err = "blob truncated";
        return false;
    }
    out = bnm_model();
    out.kind = h.kind;
    const uint8_t *b = (const uint8_t *)blob;
    for (uint32_t i = 0; i < h.n_layers; i++) {
        BlobLayer bl;
        std::memcpy(&bl, b + sizeof(BlobHead) + i * sizeof(BlobLayer), sizeof bl);
        if ((size_t)bl.offset + bl.bytes > len) { err = "blob layer payload out of range"; return false; }
        BnmLayer L;
        L.info = bl.info;
        L.weights.assign(b + bl.offset, b + bl.offset + bl.bytes);
        if ((uint64_t)L.info.weight_count * L.info.weight_elem_bytes != bl.bytes) { err = "blob layer size mismatch"; return false; }
        out.layers.push_back(std::move(L));
    }
    return bnm_validate_schedule(out, err);
}
