// C ABI, the part that needs no device: error state, version, the model objects (text parser, blob reader / writer, accessors).
// No HIP header is included here, so this file and bnm_model.cpp build with a plain host compiler: tests/test_host_sanitized.py
// compiles both with g++ -fsanitize=address,undefined and drives them with mutated headers and corrupted blobs (the blob reader
// is what every rank runs on bytes that arrived over RCCL).
#include <cstring>
#include "bnm_capi_error.hpp"
#include "bnm_model.hpp"

namespace bnm_internal {
thread_local std::string g_err;
}
using namespace bnm_internal;

extern "C" {

const char *bnm_last_error(void) { return g_err.c_str(); }
const char *bnm_version(void) { return "bitnetmcu_hip 0.1 (gfx950)"; }

int bnm_model_from_header_text(const char *text, size_t len, bnm_model **out) {
    if (!text || !out) return fail(BNM_EINVAL, "null argument");
    bnm_model *m = new bnm_model();
    std::string err;
    if (!bnm_parse_header_text(text, len, *m, err)) {
        delete m;
        return fail(err.find("support") != std::string::npos ? BNM_EUNSUPPORTED : BNM_EPARSE, err);
    }
    *out = m;
    return BNM_OK;
}

int bnm_model_from_blob(const void *blob, size_t len, bnm_model **out) {
    if (!blob || !out) return fail(BNM_EINVAL, "null argument");
    bnm_model *m = new bnm_model();
    std::string err;
    if (!bnm_deserialize(blob, len, *m, err)) {
        delete m;
        return fail(BNM_EPARSE, err);
    }
    *out = m;
    return BNM_OK;
}

size_t bnm_model_blob_size(const bnm_model *m) { return m ? bnm_serialize(*m).size() : 0; }

int bnm_model_to_blob(const bnm_model *m, void *dst, size_t cap) {
    if (!m || !dst) return fail(BNM_EINVAL, "null argument");
    std::vector<uint8_t> b = bnm_serialize(*m);
    if (cap < b.size()) return fail(BNM_EINVAL, "destination too small");
    std::memcpy(dst, b.data(), b.size());
    return BNM_OK;
}

void bnm_model_free(bnm_model *m) { delete m; }
uint32_t bnm_model_kind(const bnm_model *m) { return m ? m->kind : 0; }
uint32_t bnm_model_num_layers(const bnm_model *m) { return m ? (uint32_t)m->layers.size() : 0; }
uint32_t bnm_model_num_classes(const bnm_model *m) { return m ? m->num_classes() : 0; }
uint32_t bnm_model_input_bytes(const bnm_model *) { return 256; }

int bnm_model_layer(const bnm_model *m, uint32_t i, bnm_layer_info *info) {
    if (!m || !info || i >= m->layers.size()) return fail(BNM_EINVAL, "layer index out of range");
    *info = m->layers[i].info;
    return BNM_OK;
}

const void *bnm_model_layer_weights(const bnm_model *m, uint32_t i) {
    if (!m || i >= m->layers.size()) return nullptr;
    return m->layers[i].weights.data();
}

}  // extern "C"
