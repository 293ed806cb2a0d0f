/* seq_align.h -- version macros (mirror of reference src/seq_align.h:13-14). */
#ifndef SEQ_ALIGN_H_SEEN
#define SEQ_ALIGN_H_SEEN
#define SEQ_ALIGN_VERSION_STR "1.0.0"
#define SEQ_ALIGN_VERSION 0x100
#endif
