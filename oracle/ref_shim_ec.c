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

/* The trim pass of `bfc -1` as the reference runs it (round 6: the golden of config c5's query pass): worker_ec (correct.c:532-569, static)
 * is called on bseq1_t records built from the stream, through the reference's own kt_for as bfc_ec_cb does (correct.c:585-587), with an
 * ec_shared_t whose opt says filter_mode: max_streak (correct.c:478-497) and the keep rule (correct.c:557-567) are the reference's code,
 * nothing is restated.  worker_ec moves the kept window to the front of the read; where it began is read back from a quality string that
 * holds each position's own index (two bytes per position would be needed beyond 255 bases: such reads take max_streak a second time).
 * out_start[r] = -1 for a dropped read (s->aux = 1), else [start, end) of the window in the read. */
typedef struct { ec_step_t step; int k; const bfc_bf_t *bf; } ref_trim_t;
void ref_trim_batch(const bfc_bf_t *bf, int k, float min_frac, const uint8_t *seq, const uint64_t *off, uint64_t n_reads, int n_threads,
                    int32_t *out_start, int32_t *out_end)
{
	bfc_opt_t opt;
	ec_shared_t es;
	ec_step_t step;
	uint64_t r;
	memset(&opt, 0, sizeof(opt)); memset(&es, 0, sizeof(es));
	opt.k = k; opt.filter_mode = 1; opt.min_frac = min_frac; opt.n_threads = n_threads;
	es.opt = &opt; es.bf = bf;
	step.n_seqs = (int)n_reads; step.es = &es;
	step.seqs = (bseq1_t*)calloc(n_reads, sizeof(bseq1_t));
	for (r = 0; r < n_reads; ++r) {
		bseq1_t *s = &step.seqs[r];
		int i, l = (int)(off[r+1] - off[r]);
		s->l_seq = l;
		s->seq = (char*)malloc(l + 1); memcpy(s->seq, seq + off[r], l); s->seq[l] = 0;
		s->qual = (char*)malloc(l + 1);
		for (i = 0; i < l; ++i) s->qual[i] = (char)(i & 0xff);
		s->qual[l] = 0;
	}
	kt_for(n_threads, worker_ec, &step, (long)n_reads);
	for (r = 0; r < n_reads; ++r) {
		bseq1_t *s = &step.seqs[r];
		int l = (int)(off[r+1] - off[r]);
		if (s->aux) out_start[r] = out_end[r] = -1;
		else {
			int start;
			if (l <= 255) start = (uint8_t)s->qual[0];
			else { /* the window's start from max_streak itself on the untouched bases */
				bseq1_t t; uint64_t max;
				memset(&t, 0, sizeof(t)); t.l_seq = l; t.seq = (char*)seq + off[r];
				max = max_streak(k, bf, &t);
				start = (int)(uint32_t)max - (k - 1);
			}
			out_start[r] = start; out_end[r] = start + s->l_seq;
		}
		free(s->seq); free(s->qual);
	}
	free(step.seqs);
}
