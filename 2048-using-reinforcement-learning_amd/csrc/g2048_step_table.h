// g2048_step_table.h -- the constants the step kernels keep in LDS, and how a wavefront gets them there.
//
// 128 words, two per lane of a wavefront: one 8-byte load + one 8-byte LDS write per lane fills the whole table (every wavefront of
// a block writes the same values; a wavefront's own write precedes its reads in its LDS queue, so no block barrier is needed):
//   dir      the selector words of the direction network (g2048_board.h "direction by table"): 8 per action
//   tenths   (double)c * 0.1, c = 0..7: the ordered-pair terms of the reward (environment/game_2048.py:267-275)
//   crowded  the `empty_after <= 2` term (:263-264) as an addend: -2.0 for 0..2 empty cells, +0.0 above
//   obs      float32(code) / float32(15) (PPOAgent.normalize_state, agents/ppo_agent.py:184-195), 32 entries (codes are <= 17)
// The f64 / f32 entries are the IEEE products and quotients themselves, evaluated by the host compiler. Reading a constant from LDS
// costs an address (at most one shift) and an LDS instruction -- which issues beside the VALU instead of on it, and VALU issue is
// what bounds these kernels: 4 + 2 VALU instructions fewer per board-step than converting and multiplying (profiles/r05_kernel_heads.txt).
// Include inside the translation unit's anonymous namespace, after g2048_board.h.
#pragma once

constexpr int kStepTableWords = 128;
struct alignas(16) StepTable {
    uint32_t dir[G2048_DIR_TABLE_WORDS];
    double tenths[G2048_TENTHS];
    double crowded[G2048_CROWDED];
    float obs[32];
    uint32_t pad[kStepTableWords - G2048_DIR_TABLE_WORDS - 2 * G2048_TENTHS - 2 * G2048_CROWDED - 32];
};
static_assert(sizeof(StepTable) == 4 * kStepTableWords && offsetof(StepTable, tenths) == 128 && offsetof(StepTable, crowded) == 192 &&
              offsetof(StepTable, obs) == 328, "two words per lane");
#define G2048_OBS_INIT { 0.0f / 15.0f, 1.0f / 15.0f, 2.0f / 15.0f, 3.0f / 15.0f, 4.0f / 15.0f, 5.0f / 15.0f, 6.0f / 15.0f, 7.0f / 15.0f, \
                         8.0f / 15.0f, 9.0f / 15.0f, 10.0f / 15.0f, 11.0f / 15.0f, 12.0f / 15.0f, 13.0f / 15.0f, 14.0f / 15.0f, 15.0f / 15.0f, \
                         16.0f / 15.0f, 17.0f / 15.0f, 18.0f / 15.0f, 19.0f / 15.0f, 20.0f / 15.0f, 21.0f / 15.0f, 22.0f / 15.0f, 23.0f / 15.0f, \
                         24.0f / 15.0f, 25.0f / 15.0f, 26.0f / 15.0f, 27.0f / 15.0f, 28.0f / 15.0f, 29.0f / 15.0f, 30.0f / 15.0f, 31.0f / 15.0f }
__device__ const StepTable kStepTable = {G2048_DIR_TABLE_INIT, G2048_TENTHS_INIT, G2048_CROWDED_INIT, G2048_OBS_INIT, {}};

// reward_env_folded's constants from the LDS copy (g2048_board.h TenthByProduct is the arithmetic form)
struct TenthFromLds {
    static constexpr uint32_t kColShift = 4u;       // column counts arrive as byte offsets (count * 8)
    const StepTable *t;                             // the LDS copy
    __device__ __forceinline__ double operator()(uint32_t rows, uint32_t cols8) const
    {
        return *reinterpret_cast<const double *>(reinterpret_cast<const char *>(t->tenths) + ((rows << 3) + cols8));
    }
    __device__ __forceinline__ double crowded(double r, uint32_t empty_after) const { return r + t->crowded[empty_after]; }
};

// The fill in two halves, so that a kernel can put its own loads BETWEEN the table words' load and their LDS write: the write has
// to wait for the words, and loads issued only after it would start a second memory round trip at the head of every wavefront
// (the compiler does not move a global load up across the write). Every lane of the wavefront must run both halves.
__device__ __forceinline__ uint2 step_table_word()
{
    return reinterpret_cast<const uint2 *>(&kStepTable)[threadIdx.x & 63u];
}

__device__ __forceinline__ void step_table_store(uint4 *s_tab, uint2 word)
{
    reinterpret_cast<uint2 *>(s_tab)[threadIdx.x & 63u] = word;
}
