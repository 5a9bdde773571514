// translation unit: the H = 256 register-persistent chain (k_reni_wide256, reni_dev_wide.inc) + its host launcher
#define RENI_TU_WIDE
#include "reni_device.inc"
