/* bfc_host.h -- internal: raw access to the host-side count table for the exporter */
#ifndef BFC_HOST_H
#define BFC_HOST_H
#include "bfc_gpu.h"
#ifdef __cplusplus
extern "C" {
#endif
bfc_ch_t *bfc_ch_alloc_raw(int k, int l_pre_clamped, int cshift);
bfc_bf_t *bfc_bf_alloc_raw(int n_shift, int n_hashes); /* bfc_bf_init without the zeroing: for a filter about to be overwritten by a D2H copy */
uint64_t *bfc_ch_raw_slots(bfc_ch_t *ch);
void bfc_ch_raw_recount(bfc_ch_t *ch);
void bfc_ch_raw_set_count(bfc_ch_t *ch, uint64_t n_keys);
int bfc_ch_raw_cshift(const bfc_ch_t *ch); /* log2 of a sub-table's slot count */
/* attach order stamps (first[slots], sub_last[2^l_pre]); returns their buffers to fill */
int bfc_ch_raw_order(bfc_ch_t *ch, uint64_t **first, uint64_t **sub_last);
#ifdef __cplusplus
}
#endif
#endif
