// translation unit of libreni_hip.so -- see the header of reni_device.inc
#define RENI_TU_TRAIN_FILM 1
// (the shortened MFMA tail pads of reni_dev_train.inc are proven per translation unit on the emitted instruction streams: the FiLM
// instances interleave other fillers behind the forward GEMMs' first accumulator -- tests/isa_audit.py finds 10 / 11 states there
// with the concat instances' pad of four, twelve with six -- and take the concat values at the other three sites)
#ifndef RENI_TAIL_RB1  // (-D on the command line: the same-box A/B of profiles/tools/gpu_variants.sh)
#define RENI_TAIL_RB1 5
#endif
#include "reni_device.inc"
