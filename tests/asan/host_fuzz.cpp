// Test-only driver: the device-free part of the C ABI - bnm_model.cpp (header-text parser, blob reader /
// writer) behind bnm_capi_model.cpp's entry points - built with g++ -fsanitize=address,undefined and fed mutated headers and
// corrupted blobs.  Every input must end in a model or in a BNM_E* code; the sanitizers turn any out-of-bounds access, use after
// free, leak, signed overflow or misaligned access on the way into a non-zero exit.  tests/test_host_sanitized.py builds and runs it.
//
//   host_fuzz <iterations> <header file>... -- <blob file>...
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "../../include/bitnetmcu_hip.h"

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rnd() {      // splitmix64
    uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static size_t below(size_t n) { return n ? (size_t)(rnd() % n) : 0; }

static std::vector<uint8_t> slurp(const char *path) {
    FILE *f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(2); }
    std::vector<uint8_t> v;
    uint8_t buf[65536];
    size_t k;
    while ((k = fread(buf, 1, sizeof buf, f)) > 0) v.insert(v.end(), buf, buf + k);
    fclose(f);
    return v;
}

// walk everything a caller can reach from a parsed model: layer table, weight bytes, blob round trip
static uint64_t exercise(bnm_model *m) {
    uint64_t sum = bnm_model_kind(m) + bnm_model_num_classes(m) + bnm_model_input_bytes(m);
    const uint32_t nl = bnm_model_num_layers(m);
    for (uint32_t i = 0; i < nl + 1; i++) {      // (one past the end: must be refused)
        bnm_layer_info info;
        if (bnm_model_layer(m, i, &info) != BNM_OK) continue;
        const uint8_t *w = (const uint8_t *)bnm_model_layer_weights(m, i);
        const uint64_t bytes = (uint64_t)info.weight_count * info.weight_elem_bytes;
        for (uint64_t b = 0; b < bytes; b += 97) sum += w[b];
        if (bytes) sum += w[bytes - 1];
    }
    const size_t need = bnm_model_blob_size(m);
    std::vector<uint8_t> blob(need);
    if (bnm_model_to_blob(m, blob.data(), need ? need - 1 : 0) == BNM_OK && need) { fprintf(stderr, "short destination accepted\n"); exit(3); }
    if (bnm_model_to_blob(m, blob.data(), need) != BNM_OK) { fprintf(stderr, "to_blob failed: %s\n", bnm_last_error()); exit(3); }
    bnm_model *again = nullptr;
    if (bnm_model_from_blob(blob.data(), blob.size(), &again) != BNM_OK) { fprintf(stderr, "own blob refused: %s\n", bnm_last_error()); exit(3); }
    if (bnm_model_blob_size(again) != need) { fprintf(stderr, "blob round trip changed size\n"); exit(3); }
    bnm_model_free(again);
    return sum;
}

int main(int argc, char **argv) {
    if (argc < 4) { fprintf(stderr, "usage: host_fuzz <iterations> <header>... -- <blob>...\n"); return 2; }
    const long iters = atol(argv[1]);
    std::vector<std::vector<uint8_t>> headers, blobs;
    bool after = false;
    for (int i = 2; i < argc; i++) {
        if (!strcmp(argv[i], "--")) { after = true; continue; }
        (after ? blobs : headers).push_back(slurp(argv[i]));
    }
    uint64_t sink = 0;
    long h_ok = 0, h_bad = 0, b_ok = 0, b_bad = 0;
    // the seeds themselves must parse
    for (auto &h : headers) {
        bnm_model *m = nullptr;
        if (bnm_model_from_header_text((const char *)h.data(), h.size(), &m) != BNM_OK) { fprintf(stderr, "seed header refused: %s\n", bnm_last_error()); return 3; }
        sink += exercise(m);
        bnm_model_free(m);
    }
    for (auto &b : blobs) {
        bnm_model *m = nullptr;
        if (bnm_model_from_blob(b.data(), b.size(), &m) != BNM_OK) { fprintf(stderr, "seed blob refused: %s\n", bnm_last_error()); return 3; }
        sink += exercise(m);
        bnm_model_free(m);
    }
    for (long it = 0; it < iters; it++) {
        // ---- a mutated header: truncation / byte flips / a dropped line / a duplicated span / digits and braces swapped ----
        std::vector<uint8_t> s = headers[it % headers.size()];
        switch (it % 6) {
            case 0: s.resize(below(s.size() + 1)); break;
            case 1: for (size_t k = 1 + below(8); k--;) s[below(s.size())] = (uint8_t)rnd(); break;
            case 2: {
                size_t a = below(s.size()), e = a;
                while (a > 0 && s[a - 1] != '\n') a--;
                while (e < s.size() && s[e] != '\n') e++;
                s.erase(s.begin() + a, s.begin() + e);
                break;
            }
            case 3: {
                size_t a = below(s.size()), k = 1 + below(200);
                if (a + k > s.size()) k = s.size() - a;
                std::vector<uint8_t> span(s.begin() + a, s.begin() + a + k);
                s.insert(s.begin() + a, span.begin(), span.end());
                break;
            }
            case 4: {      // numbers in the #define lines: huge, zero, negative
                static const char *vals[] = {"0", "4294967295", "18446744073709551616", "-1", "1000000", "3", "0x7fffffff", "65"};
                const char *needle = "#define";
                std::string t((const char *)s.data(), s.size());
                size_t pos = 0, hits = 0;
                while ((pos = t.find(needle, pos)) != std::string::npos) { hits++; pos += 7; }
                if (hits) {
                    size_t pick = below(hits);
                    pos = 0;
                    for (size_t k = 0; k <= pick; k++) pos = t.find(needle, pos) + 7;
                    size_t eol = t.find('\n', pos);
                    if (eol == std::string::npos) eol = t.size();
                    size_t last_sp = t.rfind(' ', eol);
                    if (last_sp != std::string::npos && last_sp >= pos) t.replace(last_sp + 1, eol - last_sp - 1, vals[below(8)]);
                }
                s.assign(t.begin(), t.end());
                break;
            }
            default: for (size_t k = 1 + below(4); k--;) { size_t p = below(s.size()); s[p] = "{},;0x/*\n"[below(9)]; } break;
        }
        bnm_model *m = nullptr;
        if (bnm_model_from_header_text((const char *)s.data(), s.size(), &m) == BNM_OK) {
            sink += exercise(m);
            bnm_model_free(m);
            h_ok++;
        } else {
            sink += strlen(bnm_last_error());
            h_bad++;
        }
        // ---- a corrupted blob: bytes flipped in the header / layer-table region or anywhere, fields set to extremes, truncated ----
        std::vector<uint8_t> b = blobs[it % blobs.size()];
        switch (it % 5) {
            case 0: for (size_t k = 1 + below(6); k--;) b[below(b.size() < 400 ? b.size() : 400)] = (uint8_t)rnd(); break;
            case 1: for (size_t k = 1 + below(6); k--;) b[below(b.size())] = (uint8_t)rnd(); break;
            case 2: b.resize(below(b.size() + 1)); break;
            case 3: {      // a 32-bit field of the first 1 KiB set to an extreme
                static const uint32_t ext[] = {0u, 1u, 0xFFFFFFFFu, 0x7FFFFFFFu, 0x80000000u, 0x10000u, 255u, 64u};
                size_t p = below((b.size() < 1024 ? b.size() : 1024) / 4) * 4;
                uint32_t v = ext[below(8)];
                if (p + 4 <= b.size()) memcpy(b.data() + p, &v, 4);
                break;
            }
            default: {     // truncated AND corrupted
                for (size_t k = 1 + below(3); k--;) b[below(b.size() < 400 ? b.size() : 400)] = (uint8_t)rnd();
                b.resize(below(b.size() + 1));
                break;
            }
        }
        m = nullptr;
        if (bnm_model_from_blob(b.data(), b.size(), &m) == BNM_OK) {
            sink += exercise(m);
            bnm_model_free(m);
            b_ok++;
        } else {
            sink += strlen(bnm_last_error());
            b_bad++;
        }
    }
    // null / empty arguments
    bnm_model *m = nullptr;
    if (bnm_model_from_header_text(nullptr, 0, &m) == BNM_OK || bnm_model_from_blob(nullptr, 0, &m) == BNM_OK ||
        bnm_model_from_header_text("", 0, &m) == BNM_OK || bnm_model_from_blob("", 0, &m) == BNM_OK) { fprintf(stderr, "empty input accepted\n"); return 3; }
    bnm_model_free(nullptr);
    printf("{\"iterations\": %ld, \"headers_parsed\": %ld, \"headers_refused\": %ld, \"blobs_parsed\": %ld, \"blobs_refused\": %ld, \"sink\": %llu}\n",
           iters, h_ok, h_bad, b_ok, b_bad, (unsigned long long)(sink & 0xffff));
    return 0;
}
