// translation unit of libreni_hip.so -- see the header of reni_device.inc
#define RENI_TU_TRAIN_FILM 1
#include "reni_device.inc"
