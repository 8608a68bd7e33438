/* orc_reservoir.h -- ORACLE (test infrastructure, not product code).
 *
 * Literal CPU restatement of /root/reference/renderer/reservoir.py (ReSTIR-PT sample,
 * reservoir, packed storage record).
 */
#ifndef ORC_RESERVOIR_H
#define ORC_RESERVOIR_H

#include "orc_math.h"

namespace orc {

/* reservoir.py:8-19: packed record (f16 fields kept as binary16 codes) */
struct StorageReservoir {
    uint16_t M, W;
    V3 F;
    V3 rc_pos;
    uint32_t rc_normal_and_NEE_dir;
    uint16_t rc_incident_dir[2];
    V3 rc_incident_L;
    uint32_t rc_mat_info;
    uint16_t cached_jacobian_term;
    int8_t lobes;
};

/* reservoir.py:22-38 */
struct Sample {
    V3 F, rc_pos, rc_normal, rc_incident_dir, rc_incident_L, rc_NEE_dir;
    uint32_t rc_mat_info;
    float cached_jacobian_term;
    int lobes;
};

/* reservoir.py:40-141 */
struct Reservoir {
    Sample z;
    float M, weight;

    void init() { /* :46-57 */
        z.F = v3(0.0f); z.rc_pos = v3(0.0f); z.rc_normal = v3(0.0f); z.rc_incident_dir = v3(0.0f);
        z.rc_incident_L = v3(0.0f); z.rc_NEE_dir = v3(0.0f); z.rc_mat_info = 0u;
        z.cached_jacobian_term = 1.0f; z.lobes = 0;
        M = 0.0f;
        weight = 0.0f;
    }
    void update_cached_jacobian_term(V3 x1) { /* :59-62 */
        V3 d = z.rc_pos - x1;
        z.cached_jacobian_term = dot(d, d) / dm_abs(dot(normalized(d), z.rc_normal));
    }
    bool input_sample(float in_w, const Sample& in_z, dm_rng* rng, bool force_add = false) { /* :64-74 */
        M += 1.0f;
        bool selected = false;
        if (in_w > 0.0f) {
            weight += in_w;
            bool lt = dm_rng_f32(rng) * weight <= in_w;
            selected = lt || force_add;
            if (selected) z = in_z;
        }
        return selected;
    }
    bool merge(const Reservoir& in_r, float in_w, dm_rng* rng, bool force_add = false) { /* :76-86 */
        M += in_r.M;
        bool selected = false;
        if (in_w > 0.0f) {
            weight += in_w;
            bool lt = dm_rng_f32(rng) * weight <= in_w;
            selected = lt || force_add;
            if (selected) z = in_r.z;
        }
        return selected;
    }
    void finalize_without_M() { /* :96-102 */
        float p_hat = luminance(z.F);
        if (p_hat < 1e-6f) weight = 0.0f;
        else weight = weight / p_hat;
    }
    StorageReservoir encode() const { /* :104-124 */
        StorageReservoir enc;
        enc.M = dm_f32_to_f16(M);
        enc.W = dm_f32_to_f16(weight);
        enc.F = z.F;
        enc.rc_pos = z.rc_pos;
        uint16_t on[2], od[2];
        encode_unit_vector_3x16(z.rc_normal, on);
        encode_unit_vector_3x16(z.rc_NEE_dir, od);
        enc.rc_normal_and_NEE_dir = encode_u32_arb8(dm_f16_to_f32(on[0]), dm_f16_to_f32(on[1]), dm_f16_to_f32(od[0]),
                                                    dm_f16_to_f32(od[1]));
        encode_unit_vector_3x16(z.rc_incident_dir, enc.rc_incident_dir);
        enc.rc_incident_L = z.rc_incident_L;
        enc.rc_mat_info = z.rc_mat_info;
        enc.cached_jacobian_term = dm_f32_to_f16(z.cached_jacobian_term);
        enc.lobes = (int8_t)z.lobes;
        return enc;
    }
    void decode(const StorageReservoir& enc) { /* :126-141 */
        M = dm_f16_to_f32(enc.M);
        weight = dm_f16_to_f32(enc.W);
        z.F = enc.F;
        z.rc_pos = enc.rc_pos;
        V4 data = decode_u32_arb8(enc.rc_normal_and_NEE_dir);
        z.rc_normal = decode_unit_vector_3x16(data.x, data.y);
        z.rc_NEE_dir = decode_unit_vector_3x16(data.z, data.w);
        z.rc_incident_dir = decode_unit_vector_3x16(enc.rc_incident_dir);
        z.rc_incident_L = enc.rc_incident_L;
        z.rc_mat_info = enc.rc_mat_info;
        z.cached_jacobian_term = dm_f16_to_f32(enc.cached_jacobian_term);
        z.lobes = (int)enc.lobes;
    }
};

} /* namespace orc */
#endif
