/* pesto_oracle.h - CPU restatement of the PeSTo forward pass. TEST INFRASTRUCTURE (see pesto_oracle.c). */
#ifndef PESTO_ORACLE_H
#define PESTO_ORACLE_H
#include <stdint.h>
#include "../include/pesto_hip.h"
#ifdef __cplusplus
extern "C" {
#endif
struct oracle_model;
int oracle_blob_size(const pesto_config* cfg, int64_t* n);
int oracle_create(const pesto_config* cfg, const float* weights, int64_t n_weights, struct oracle_model** out);
void oracle_destroy(struct oracle_model* m);
void oracle_set_threads(int n);   /* OpenMP threads of the following calls; n < 1: all cores */
int oracle_embed(const struct oracle_model* m, int64_t N, const float* q0, float* q_out);
int oracle_unpack(int64_t N, int k, const float* X, const int32_t* ids, int32_t* ids_s, float* D, float* R);
int oracle_layer(const struct oracle_model* m, int layer, int64_t N1, int k, const int32_t* ids_s, const float* D,
                 const float* R, float* q, float* p);
int oracle_pool(const struct oracle_model* m, int64_t N, int64_t R, const float* q, const float* p,
                const int32_t* res_of_atom, float* qr, float* pr, float* z);
/* stop_after < 0: all layers. q_state/p_state (optional): state after the executed layers, incl. sink row */
int oracle_forward(const struct oracle_model* m, int64_t N, int64_t R, int k, const float* X, const int32_t* ids,
                   const float* q0, const int32_t* res_of_atom, float* z, float* q_state, float* p_state, int stop_after);
#ifdef __cplusplus
}
#endif
#endif
