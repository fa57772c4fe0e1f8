/* gq_host_model.h - GqModelDesc -> GqDevModel / GqDevBatch lowering (host, no HIP). */
#pragma once
#include <cstddef>
#include <vector>

#include "gq.h"
#include "gq_model_dev.h"

int gq_build_dev_model(const GqModelDesc* d, GqDevModel* out, std::vector<float>* vx, std::vector<float>* vy,
                       std::vector<float>* vz, char* err, size_t errlen);
/* elevations of the height field in metres relative to hfield_pos[2] ([nrow][ncol]); the caller places them in memory the
 * kernels can read and points GqDevModel::hf_data at it */
void gq_hfield_heights(const GqModelDesc* d, std::vector<float>* out);
int gq_build_dev_batch(int n_envs, const int32_t* obs_ids, int n_obs, const int32_t* legs_order, GqDevBatch* out,
                       char* err, size_t errlen);
int gq_obs_dim_host(int id);
void gq_fill_imu(GqDevBatch* b, const GqImuCfg* cfg);
