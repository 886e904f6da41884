/*
 * freddie_host.h -- C-ABI of the native host side of the segmentation stage:
 * multi-threaded parsing of split_*.tsv / reads_*.tsv into the flat arrays fseg_upload() takes, and
 * multi-threaded gaps / poly-A annotation + segment_*.tsv writing from the device results.
 *
 * It replaces, for the batched CLI path, the reference's Python
 *   read_split()      py/freddie_segment.py:121-171  (incl. the read_reps grouping :165-170)
 *   read_sequence()   py/freddie_segment.py:174-185
 *   get_unaligned_gaps_and_polyA() and helpers   py/freddie_segment.py:289-472
 *   the writer part of run_segment()             py/freddie_segment.py:703-732
 * and (row N2) a binary side-car of a partition's two TSVs that the loader takes instead of parsing them again.
 * It does no segmentation arithmetic (that is libfreddie_seg.so's job, on the GPU).
 */
#ifndef FREDDIE_HOST_H
#define FREDDIE_HOST_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct fhost_batch fhost_batch;

/* Parse n partitions (split_paths[i], reads_paths[i]) with n_threads worker threads.
 * Returns NULL on allocation failure; otherwise a batch whose fhost_error() is "" on success. */
fhost_batch *fhost_load(const char *const *split_paths, const char *const *reads_paths, int32_t n, int32_t n_threads);
void fhost_free(fhost_batch *b);
const char *fhost_error(const fhost_batch *b);   /* "" when the batch is usable */

/* Sizes and flat arrays in exactly the layout of fseg_batch (include/freddie_seg.h).  Pointers stay valid
 * until fhost_free(). */
int32_t fhost_n_part(const fhost_batch *b);
int64_t fhost_n_reads(const fhost_batch *b);
const int64_t *fhost_part_iv_off(const fhost_batch *b);
const int32_t *fhost_iv_start(const fhost_batch *b);
const int32_t *fhost_iv_end(const fhost_batch *b);
const int64_t *fhost_part_rep_off(const fhost_batch *b);
const int32_t *fhost_rep_weight(const fhost_batch *b);
const int64_t *fhost_rep_exon_off(const fhost_batch *b);
const int32_t *fhost_ex_ts(const fhost_batch *b);
const int32_t *fhost_ex_te(const fhost_batch *b);

/* Annotate and write: for partition p the device results are final positions
 * final_pos[part_final_off[p] .. part_final_off[p+1]) and the label rows of its read reps at
 * labels[label_off[p] + rep * (F_p - 1)] (ASCII '0','1','2'; what fseg_download() returns).
 * out_paths[p] receives segment_<contig>_<tint>.tsv.  Returns 0 on success; on failure fhost_error() says
 * which reference assertion would have fired. */
int32_t fhost_write(fhost_batch *b, const int64_t *part_final_off, const int32_t *final_pos, const int64_t *label_off,
                    const uint8_t *labels, const char *const *out_paths, int32_t n_threads);

/* The same with the labels at two bits each, as fseg_results_packed() hands them over (label byte g of the arena =
 * bits 2(g & 3).. of labels2[g >> 2]; label_off still counts labels): the rows are unpacked straight into the TSV. */
int32_t fhost_write_packed(fhost_batch *b, const int64_t *part_final_off, const int32_t *final_pos, const int64_t *label_off,
                           const uint8_t *labels2, const char *const *out_paths, int32_t n_threads);

/* ---- binary side-car ------------------------------------------------------------
 * split_<contig>_<tint>.fsc, written next to the TSVs that py/freddie_split.py:445-481 produces, holds the parsed
 * form of both files (flat exon / CIGAR arrays, the read_reps grouping of py/freddie_segment.py:165-170, sequences
 * at two bits per base with an exception list for bytes other than ACGT).  A side-car is used only when the sizes
 * and mtimes recorded in it equal those of its two TSVs (and, with verify_checksum, its payload checksum holds);
 * otherwise the TSVs are parsed, so results never depend on whether side-cars exist. */

/* Like fhost_load(); sidecar_paths may be NULL, and any entry may be NULL. */
fhost_batch *fhost_load_sidecar(const char *const *split_paths, const char *const *reads_paths,
                                const char *const *sidecar_paths, int32_t n, int32_t n_threads, int32_t verify_checksum);
int32_t fhost_n_from_sidecar(const fhost_batch *b);   /* partitions of the batch that came from a side-car */

/* Write the side-car of every partition of a loaded batch (atomically: temp file + rename).  Returns 0 on success. */
int32_t fhost_sidecar_write(fhost_batch *b, const char *const *split_paths, const char *const *reads_paths,
                            const char *const *sidecar_paths, int32_t n_threads);

/* ---- the split directory's listing and the empty .log files (round 6) ---------------------------------------------------
 * What main() does before and around the batches as loops of system calls (py/freddie_segment.py:852-857: one directory per
 * contig, one split_<contig>_<tint>.tsv per partition; :695: an empty segment_*.log per partition), natively and on n_threads:
 * fhost_discover lists every split_*.tsv of every contig directory with its size (what the scatter and the batching weigh a
 * partition by); fhost_touch creates (or truncates) the given files. */
typedef struct fhost_listing fhost_listing;
fhost_listing *fhost_discover(const char *split_dir, int32_t n_threads);   /* NULL: out of memory; else check fhost_listing_error() */
void fhost_listing_free(fhost_listing *l);
const char *fhost_listing_error(const fhost_listing *l);                   /* "" when the listing is usable */
int64_t fhost_listing_n(const fhost_listing *l);                           /* partitions found */
int32_t fhost_listing_n_contigs(const fhost_listing *l);
const char *fhost_listing_contig(const fhost_listing *l, int32_t k);       /* name of contig directory k */
const int32_t *fhost_listing_contig_of(const fhost_listing *l);            /* per partition: index of its contig directory */
const int64_t *fhost_listing_tint(const fhost_listing *l);                 /* per partition: the tint id from the file name */
const int64_t *fhost_listing_size(const fhost_listing *l);                 /* per partition: bytes of its split TSV */
int32_t fhost_touch(const char *const *paths, int32_t n, int32_t n_threads);   /* 0 on success */

#ifdef __cplusplus
}
#endif
#endif
