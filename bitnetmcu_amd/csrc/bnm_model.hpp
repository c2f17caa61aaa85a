// Host-side model representation: the BitNetMCU_model.h interchange format parsed at run time.
// No HIP in this file — it is unit-tested on CPU.
//
// Reference for the format: the writer exportquant.py:49-263 and its three textual dialects in
// the tree (BitNetMCU_model_fc.h:8-25, mcu/BitNetMCU_model_12k.h:8-31, BitNetMCU_model_cnn.h:8-34).
#pragma once
#include <cstdint>
#include <string>
#include <vector>
#include "../../include/bitnetmcu_hip.h"

struct BnmLayer {
    bnm_layer_info info{};
    std::vector<uint8_t> weights;  // raw array bytes exactly as the C compiler would lay them out
};

struct bnm_model {
    uint32_t kind = BNM_KIND_FC;
    std::vector<BnmLayer> layers;

    // FC layers in schedule order (indices into layers)
    std::vector<uint32_t> fc_layers() const;
    uint32_t num_classes() const;
};

// Codec helpers shared by host validation and launch code.
// bits of one packed field for the 32-bit-word codecs, 0 for ternary/unknown
int bnm_codec_field_bits(int32_t bpw);
// does the codec decode to values that fit int8 (everything except FP130's +-128)?
bool bnm_codec_known(int32_t bpw);
// number of array elements a well-formed FC layer must carry
uint64_t bnm_fc_weight_count(int32_t bpw, uint32_t n_input, uint32_t n_output);
// activations actually consumed (ternary n_input is padded to a multiple of 10,
// exportquant.py:132-137; pad trits are zero so the image's 256 bytes are all that is read)
uint32_t bnm_fc_real_inputs(const bnm_layer_info &li, uint32_t prev_outputs);

bool bnm_parse_header_text(const char *text, size_t len, bnm_model &out, std::string &err);
std::vector<uint8_t> bnm_serialize(const bnm_model &m);
bool bnm_deserialize(const void *blob, size_t len, bnm_model &out, std::string &err);
// topology check: what BitMnistInference (BitNetMCU_MNIST_dll.c:48-121) can execute
bool bnm_validate_schedule(const bnm_model &m, std::string &err);
