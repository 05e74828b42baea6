/* ref_shim_ec.c -- TEST INFRASTRUCTURE.  Reaches bfc_ec_kcov (correct.c:96-117) of the reference: its argument type
 * ecseq_t and the converter bfc_seq_conv are private to correct.c, so this file is compiled as ONE translation unit with
 * the reference's correct.c included from where it lies (-I$(REF), see Makefile; nothing is copied).  Built into
 * oracle/_ref/libbfcref_ec.so together with the reference's other objects. */
#include "correct.c"

/* the three globals bfc.c:13-15 owns (its main() is not linked here) */
int bfc_verbose = 1;
double bfc_real_time;
bfc_kmer_t bfc_kmer_null = {{0,0,0,0}};

/* out[i] = lcov | hcov<<6 | solid_end<<12 | high_end<<13 after bfc_seq_conv + bfc_ec_kcov on one read */
void ref_kcov(const bfc_ch_t *ch, int k, int min_occ, int qthres, const char *seq, const char *qual, uint16_t *out)
{
	ecseq_t s;
	size_t i;
	kv_init(s);
	bfc_seq_conv(seq, qual, qthres, &s, 0);
	bfc_ec_kcov(k, min_occ, &s, ch);
	for (i = 0; i < s.n; ++i)
		out[i] = (uint16_t)(s.a[i].lcov | s.a[i].hcov << 6 | s.a[i].solid_end << 12 | s.a[i].high_end << 13);
	free(s.a);
}
