/* focus_debug.h -- bring-up probes exported by libfocus_amd.so for the TEST SUITE only; not part of the operator ABI
 * (include/focus_amd.h).  focus_amd/_lib.py parses this header too, so the probes are callable through ctypes. */
#pragma once
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Hardware probe (bring-up aid, used by tests only): fills an LDS image img[r][c] = 100*r + c (16 rows x 64
 * columns of int16, 128-B rows), issues ONE ds_read_b64_tr_b16 per lane with lane l of each 16-lane group g
 * addressing row 4*g + (l%16)/4, column 4*(l%4), and returns the 4 int16 each lane received: out [64][4]. */
int focus_debug_tr16_probe(int16_t* out, void* stream);

#ifdef __cplusplus
}
#endif
