/* seq_align.h -- source compatibility with noporpoise/seq-align: everything lives in
 * seqalign_compat.h (see there for the per-declaration reference citations). */
#include "seqalign_compat.h"
