/* tests/native/abi_is_c.c -- include/mms.h and include/mms_layer.h must be consumable from plain C (the
 * boundary a cgo / JNI / ctypes binding sees): compiled with `gcc -std=c99 -Wall -Werror -pedantic`. */
#include <stddef.h>
#include "mms.h"
#include "mms_layer.h"

/* take the address of one entry point per family with its declared type: a signature that needed C++
 * would not get this far */
typedef int (*fwd_t)(int, int, int, int, int, int, const float*, const float*, const float*, const float*,
                     float*, float*, float*, void*, size_t, void*);
static fwd_t use_forward(void) { return &mms_simcross_forward_f32; }

typedef void (*any_fn)(void);
/* entry points added over the round, referenced by name: a typo or a C++-only declaration fails here */
static const any_fn table[] = {
    (any_fn)mms_layer_create, (any_fn)mms_simmatrix_backward_cached_f32, (any_fn)mms_embed_simcross_forward_f32,
    (any_fn)mms_set_euclid_backward_mode, (any_fn)mms_triplet_euclid_step_f32, (any_fn)mms_rank_map_mrr_f32};

int mms_abi_is_c_probe(void) {
  return use_forward() != NULL && MMS_OK == 0 && MMS_VERSION > 0 && sizeof(table) / sizeof(table[0]) == 6;
}
