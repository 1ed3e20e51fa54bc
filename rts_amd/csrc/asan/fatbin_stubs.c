__attribute__((aligned(4096))) char __hip_fatbin_3f03275db16c7ec6[4096] = {0};
__attribute__((aligned(4096))) char __hip_fatbin_a4b1323ea19961cb[4096] = {0};
__attribute__((aligned(4096))) char __hip_fatbin_d87bff6dc9a8ea9a[4096] = {0};
__attribute__((aligned(4096))) char __hip_fatbin_dd7d75cad18b0d32[4096] = {0};
__attribute__((aligned(4096))) char __hip_fatbin_efdfe5b69200bb02[4096] = {0};
