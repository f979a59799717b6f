// bf16 instantiation of the per-layer decoder kernel: the source of decoder_layer.hip compiled with bf16 storage and
// v_mfma_f32_16x16x32_bf16 (entry point codetr_decoder_layer_bf16; see the element-type note in that file).
#define CODETR_DEC_BF16 1
#include "decoder_layer.hip"
