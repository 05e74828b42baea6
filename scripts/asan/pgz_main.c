/* The parallel inflate (bfc_pgz.h) under AddressSanitizer/UBSan: it decodes from guessed bit positions of files that may be damaged.
 *   build/asan_pgz file threads chunk window  -> "rc bytes crc32 guessed redone" (same loop as bfc_pgz_digest in bfc_count.c) */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <fcntl.h>
#include <unistd.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include "bfc_pgz.h"

int main(int argc, char **argv)
{
	struct stat st;
	int fd, rc = 0;
	uint8_t *z;
	pgz_t *g;
	uint64_t pos = 0, window;
	uint32_t crc = (uint32_t)crc32(0L, Z_NULL, 0);
	if (argc < 5) return 2;
	fd = open(argv[1], O_RDONLY);
	if (fd < 0 || fstat(fd, &st) != 0) return 3;
	z = (uint8_t*)malloc((size_t)st.st_size + 1); /* an exact heap copy: any read past the end is reported */
	if (read(fd, z, (size_t)st.st_size) != st.st_size) return 3;
	close(fd);
	window = strtoull(argv[4], 0, 10);
	g = pgz_open(z, (size_t)st.st_size, atoi(argv[2]), (size_t)strtoull(argv[3], 0, 10));
	for (;;) {
		const uint8_t *p; uint64_t avail; int eof;
		if (pgz_ensure(g, pos, window, &p, &avail, &eof) != 0) { rc = -2; break; }
		crc = (uint32_t)crc32(crc, p + pos, (uInt)(avail - pos));
		pos = avail;
		if (eof) break;
	}
	printf("%d %llu %u %llu %llu\n", rc, (unsigned long long)pos, crc, (unsigned long long)g->n_spec, (unsigned long long)g->n_redo);
	pgz_close(g);
	free(z);
	return 0;
}
