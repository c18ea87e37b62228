// textemit.h -- the text mode of the emitter (format.hip): the tuples of a prefix leave the card as the reference server's stdout
// lines (metaserver.cpp:472-484) instead of binary arrays.  The exact entropy (metaserver.cpp:366-389), the emin / emax test (:413),
// the lines' lengths, a scan and the lines themselves are computed on the device from the tuple arrays tuple_fill_kernel left there;
// what crosses the bus is the text alone.
#pragma once
#include "common.h"

namespace dsm {

struct TextEmit;   // device-side state of one emitter: tables, scratch, the text buffers (format.hip)
TextEmit* text_emit_create(int device);
void text_emit_destroy(TextEmit* t);
// Tuples [t0, t1) of a prefix (device arrays, offsets as tuple_fill_kernel uses them) -> *text (pinned host memory, valid until the
// next call), *len bytes; the tuples and pairs that passed the entropy test are counted.  Runs on the emitter's own stream and
// returns when the text is on the host.
int text_emit_chunk(TextEmit* t, u32 t0, u32 t1, const u32* path_off, const u32* pair_off, const u32* ids, const u64* freqs, const char* paths,
                    u32 d, double emin, double emax, const char** text, size_t* len, u64* kept_tuples, u64* kept_pairs);

}  // namespace dsm
