// Launchers of the kernel-2 families, one translation unit each (k_*.hip).  Each enqueues ONE launch of the
// planned variant on the context's stream, writing |p| to `pm` (the current output buffer) and the other planned
// outputs to the context's buffers; errors surface through hipGetLastError in olx_field_launch.
#pragma once
struct olx_ctx;
void olx_launch_accum(olx_ctx* c, float* pm);        // 2a  field_accum_k
void olx_launch_accum_dir(olx_ctx* c, float* pm);    // 2a-d field_accum_dir_k (piston directivity; needs c->d_tab2)
bool olx_launch_shared(olx_ctx* c, float* pm);       // 2b  field_shared_k (false: no instantiation for the planned shape)
void olx_launch_mfma(olx_ctx* c, float* pm);         // 2c  field_mfma_k
void olx_launch_lattice(olx_ctx* c, float* pm);      // 2d  field_lattice_k
void olx_launch_coset(olx_ctx* c, float* pm);        // 2e  field_coset_k
void olx_launch_cosetp(olx_ctx* c, float* pm);       // 2g  field_cosetp_k (2e's NT = 2 shape, planes in the MFMA rows)
void olx_launch_toep(olx_ctx* c, float* pm);         // 2f  field_toep_k (single steering column on a lattice array)
void olx_pack_toep(olx_ctx* c);                      //     its Toeplitz weight fragments
void olx_launch_hetero(olx_ctx* c, float* pm);       // 2h  field_hetero_k
void olx_launch_hmarch(olx_ctx* c, float* pm);       // 2m  field_hmarch_k (marched ray sums: one launch per plane segment)
void olx_pack_hetero(olx_ctx* c);                    //     its steering table (c->nf foci per launch tile)
