// translation unit of libreni_hip.so -- see the header of reni_device.inc
#define RENI_TU_FILM_BF16 1
#include "reni_device.inc"
