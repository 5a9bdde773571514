// translation unit of libreni_hip.so -- see the header of reni_device.inc
#define RENI_TU_TRAIN_FILM 1
// (the shortened MFMA tail pads of reni_dev_train.inc are proven on the concat instances' instruction streams; the FiLM instances
// interleave other fillers -- tests/isa_audit.py finds 10 / 11 states there -- and keep the full twelve)
#define RENI_TAIL_RB2 11
#define RENI_TAIL_RB1 11
#define RENI_TAIL_DWDX 11
#define RENI_TAIL_DWDX_A 11
#include "reni_device.inc"
