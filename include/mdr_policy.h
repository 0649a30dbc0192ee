/*
 * mdr_policy.h - fused policy forward + action sampling for rollout collection.
 *
 * Replaces, for all agents of all envs at once, what the reference does per agent and step on the CPU:
 *   PPO.select_action (agents/ppo.py:68-75): actor_net(state) -> Categorical(action_prob).sample() -> (action, prob)
 *   Actor.forward      (agents/network.py:14-33): Linear/ReLU stack (two hidden layers) with a softmax head, 2 actions
 *
 * One kernel: observation rows [A][F] -> layer 1 -> ReLU -> layer 2 -> ReLU -> logits -> softmax -> sample.
 * The two dense layers run on the matrix cores in exact fp32 (v_mfma_f32_32x32x2_f32: a k-ordered fp32 fma chain, no
 * reduced precision), the hidden activations never leave the registers, biases ride along as one extra input / hidden
 * unit that is constant 1.  The weights are handed over pre-arranged in MFMA fragment order (mdr_actor_pack_* below
 * describe it; mdr_amd/policy.py builds it from a torch state_dict).
 *
 * Part of libmdr_hip.so; plain C ABI, device pointers, stream-ordered, never synchronises.
 */
#ifndef MDR_POLICY_H
#define MDR_POLICY_H

#include <stdint.h>

#include "mdr.h"

#ifdef __cplusplus
extern "C" {
#endif

#define MDR_ACTOR_MAX_HIDDEN 127 /* per hidden layer: 127 units + the constant-1 unit fill four 32-row MFMA blocks */

enum mdr_actor_layout {
  MDR_ACTOR_FRAG32 = 0, /* v_mfma_f32_32x32x2_f32: 32 agents per wavefront, any num_state that fits the LDS */
  MDR_ACTOR_FRAG16 = 1, /* v_mfma_f32_16x16x4_f32: 16 agents per wavefront, hidden units padded to 112 instead of 128 rows and a
                           quarter of the accumulator registers; num_state <= 128 (16 feature registers per lane up to 64
                           features, 32 beyond - observations with the optional message columns, utils.py:858-866) */
  MDR_ACTOR_BF16X3 = 2, /* v_mfma_f32_16x16x32_bf16 on operands split into bf16 head + tail (x = xh + xl): w x ~ wh xh + wl xh +
                           wh xl, fp32 accumulation - 16 significand bits per operand instead of 24 (probabilities within ~1e-5
                           of the fp32 forward) at 16 / 3 times the fp32 matrix rate; num_state <= 128 (mdr_actor_sample; 64 through
                           mdr_env_actor_sample).  frag1 / frag2 hold bf16 */
  MDR_ACTOR_FRAG16T = 3 /* MDR_ACTOR_FRAG16 for hidden layers of 97..100 units (the reference's [100, 100]): six 16-row blocks on
                           v_mfma_f32_16x16x4_f32 and the last 1..4 units of either layer on v_mfma_f32_4x4x1_16B_f32 (exact fp32,
                           a third of the time of the block it replaces): ~9 % fewer matrix cycles per agent */
};

typedef struct mdr_actor {
  uint32_t struct_size;
  int32_t layout;      /* mdr_actor_layout: how frag1 / frag2 / wdiff are arranged */
  int32_t num_state;   /* F: floats per observation row */
  int32_t hidden1;     /* units of hidden layer 1 (<= MDR_ACTOR_MAX_HIDDEN) */
  int32_t hidden2;     /* units of hidden layer 2 (<= MDR_ACTOR_MAX_HIDDEN) */
  int32_t greedy;      /* 0: action ~ Categorical(softmax) (PPOAgent.act, agents/rl_controllers.py:28-36); 1: action = argmax of the two
                          outputs (DQNAgent.act, rl_controllers.py:53-60: the same Linear/ReLU stack read as Q-values), no draw */
  int32_t feature_order; /* which input feature column k of W1 multiplies: 0 = MDR_FEATURES_NORMSTATE, normStateDict's own order
                            (mdr_actor_sample); 1 = MDR_FEATURES_OBSERVE, the order mdr_env_actor_sample stages the observation
                            in - the messages first (k = 4 m + field), then the own features in normStateDict order (k = M + i,
                            M = observe_msg_floats): W1's columns permuted with k -> normStateDict index (k < M ? own + k : k - M),
                            own = num_state - M */
  int32_t observe_msg_floats; /* MDR_FEATURES_OBSERVE: M = 4 * nb_comm message floats lead the staged row (40 for the reference's
                                 default 10 neighbours; 0 = no messages) */
  /* device, MFMA fragment order.  W1z / W2z / W3z: the weight matrices zero-padded to 128 rows / columns.
   * MDR_ACTOR_FRAG32 (S1 = ceil((F + 1) / 2), S2 = mdr_actor_steps2, r = lane & 31, h = lane >> 5) carries the biases as a
   * constant-1 input feature / hidden unit:  W1e = [[W1 b1] [0 1]],  W2e = [[W2 b2] [0 1]],  W3e = [W3 b3]:
   *   frag1[s][lane][mb < 4]  = W1e[32 mb + r][h S1 + s]
   *   frag2[q][lane][mb < 4]  = W2e[32 mb + r][k2],  k2 = 32 (q >> 4) + (q & 3) + 8 ((q >> 2) & 3) + 4 h  (the accumulator row the lane holds)
   *   wdiff[mb < 4][reg < 16][h] = W3e[0][row] - W3e[1][row],  row = 32 mb + (reg & 3) + 8 (reg >> 2) + 4 h      (128 floats)
   * MDR_ACTOR_FRAG16 (S1 = ceil(F / 4), S2 = 4 floor(H1 / 16) + ceil((H1 % 16) / 4), r = lane & 15, g = lane >> 4); the biases start
   * the accumulators.  Hidden-1 units of a partial last 16-row block are stored transposed - unit 16 b + j in row (of W1z, b1) /
   * column (of W2z) 16 b + 4 (j % 4) + j / 4 - so that the block's first ceil((H1 % 16) / 4) k-steps of layer 2 carry them all:
   *   frag1[s][lane][mb < 8]  = W1z[16 mb + r][g S1 + s]
   *   frag2[q][lane][mb < 8]  = W2z[16 mb + r][16 (q >> 2) + 4 g + (q & 3)]
   *   wdiff (388 floats) = d[mb < 8][reg < 4][g < 4] | b1[mb][g][reg] | b2[mb][g][reg] | b3[0] - b3[1] | 0 0 0,
   *                        d = W3z[0][row] - W3z[1][row],  b1 / b2 at row (0 past H),  row = 16 mb + 4 g + reg
   * MDR_ACTOR_FRAG16T: as MDR_ACTOR_FRAG16 with S2 = 25, except slot mb = 6 of every fragment and the tail's biases / head weights,
   * u = 96 + (lane & 3) (zero past the layer's units):
   *   frag1[s][lane][6] = W1[u][g S1 + s]      frag2[q < 24][lane][6] = W2[u][16 (q >> 2) + 4 g + (q & 3)]
   *   frag2[24][lane][mb < 6] = W2z[16 mb + r][96 + g]      frag2[24][lane][6] = W2[u][96 + g]
   *   d[6][reg][g] = (g == 0) (W3[0][96 + reg] - W3[1][96 + reg]),  b1 / b2 [6][g][reg] = (g == 0) b[96 + reg]
   * MDR_ACTOR_BF16X3 (S1 = ceil(F / 32), S2 = 4, r = lane & 15, g = lane >> 4, fragments of 8 bf16, t = 0 head / 1 tail):
   *   frag1[s][mb < 8][t][lane][j < 8] = split_t(W1z[16 mb + r][(4 s + g) 8 + j])
   *   frag2[s][mb < 8][t][lane][j < 8] = split_t(W2z[16 mb + r][16 (2 s + (j >> 2)) + 4 g + (j & 3)])
   *   wdiff as MDR_ACTOR_FRAG16;  split_0(w) = bf16(w), split_1(w) = bf16(w - split_0(w)), round to nearest even */
  const void *frag1;
  const void *frag2;
  const float *wdiff;  /* 128 floats (FRAG32) or 388 */
} mdr_actor_t;

int64_t mdr_actor_steps1(int32_t layout, int32_t num_state);        /* S1 */
/* S1 of an actor packed with feature_order = 1 (MDR_FEATURES_OBSERVE) for the exact-fp32 layouts: the k-steps its observe -> act
 * kernel is compiled for - 13 (num_state <= 52, the default 51 included), 15 (<= 60) or 16 (<= 64); W1 columns past num_state are
 * zeros.  Every other (layout, order): mdr_actor_steps1.  frag1 then holds S1 * (mdr_actor_frag1_floats / mdr_actor_steps1) units. */
int64_t mdr_actor_steps1_order(int32_t layout, int32_t num_state, int32_t feature_order);
int64_t mdr_actor_steps2(int32_t layout, int32_t hidden1);          /* S2 */
int64_t mdr_actor_frag1_floats(int32_t layout, int32_t num_state);  /* size of frag1 in 4-byte units */
int64_t mdr_actor_frag2_floats(int32_t layout, int32_t hidden1);    /* size of frag2 in 4-byte units */

/* For every agent a < nb_agents: probs = softmax(actor(obs[a])), u = Philox4x32-10(key = seed, counter = (a, step, stream))
 * uniform in (0,1), action = u < probs[0] ? 0 : 1  (Categorical(probs).sample()), a_prob = probs[action].
 * `obs`: observation rows [nb_agents][F] when obs_plane_stride == 0 (mdr_env_obs_vector MDR_OBS_ROWS), or feature planes
 * [F][obs_plane_stride] with obs_plane_stride >= nb_agents (MDR_OBS_PLANES: the lanes of a wavefront then read consecutive
 * floats instead of one cache line each).  `action` uint8 [nb_agents], `a_prob` float [nb_agents] (may be NULL), `probs`
 * float [nb_agents][2] (may be NULL).  `step_dev` (device int32, may be NULL) is added to `step` on the device - point it at
 * the time index of mdr_buffers_t.cursor ([1]) so that a captured launch draws fresh numbers at every replay.  Returns 0, or -1 (invalid argument) / -3 (HIP error) / -4 (shape without a kernel). */
int mdr_actor_sample(const mdr_actor_t *actor, const float *obs, int64_t obs_plane_stride, int64_t nb_agents, uint64_t seed,
                     uint64_t step, const int32_t *step_dev, uint8_t *action, float *a_prob, float *probs, void *stream);

/* Observe -> act in ONE kernel (SURVEY 8f-1 + 8f-2 fused): what train_ppo.py:69-75 does per agent - utils.normStateDict(obs_dict[i]) then
 * PPO.select_action - for every agent of every env, WITHOUT materialising the 204-byte observation rows: each wavefront stages the
 * compact state of its 32 (16) consecutive houses and their 5 + 5 circular neighbours in LDS (own features and SingleHouse.message
 * records, the very arithmetic of mdr_env_obs_vector), and the matrix-core forward reads its B operand from there.  Draws, outputs and
 * `step_dev` as mdr_actor_sample (agent index = env * nb_houses + house).  Covers the reference's DEFAULT observation only - every
 * optional state / message column off, agents_comm_mode "neighbours" with nb_agents_comm = 10, no link defects (spec says which; 51
 * features, hence nb_houses >= 11) - with unsharded houses and an actor packed as MDR_ACTOR_FRAG16 or MDR_ACTOR_BF16X3 in
 * MDR_FEATURES_OBSERVE order; anything else returns MDR_ERR_UNSUPPORTED (-4): fall back to mdr_env_obs_vector + mdr_actor_sample.
 * Any cluster size: tiles of 32 (16) consecutive agents may start anywhere in an env and span several (the reference trains with 20
 * houses and deploys with 50); nb_houses % 32 == 0 takes a leaner staging path.
 * `rows_out` (may be NULL; 16-byte aligned for the wide-store path, else 4-byte stores): the observation rows themselves, float [nb_agents][51] in normStateDict order - bit for
 * bit what mdr_env_obs_vector(MDR_OBS_ROWS) writes - copied out of the staged window on the side, for callers that keep the `state` of
 * every transition (train_ppo.py:87-98): the rows are then written once and never read back by the policy. */
int mdr_env_actor_sample(mdr_env_t *env, const mdr_obs_spec_t *spec, const mdr_actor_t *actor, uint64_t seed, uint64_t step,
                         const int32_t *step_dev, uint8_t *action, float *a_prob, float *probs, float *rows_out, void *stream);

/* The same for senders that are NOT the circular neighbours: a static link table (spec->links: agents_comm_mode closed_groups /
 * random_fixed / neighbours_2D, ClusterHouses.build_agent_comm_links env 806-902) or random_sample (spec->random_links, env 976-983:
 * nb_comm distinct senders drawn per house and step).  The call first writes every house's SingleHouse.message record into
 * `msg_scratch` (device float [nb_envs][nb_houses][4], 16-byte aligned; mdr_env_obs_messages' kernel) and - random_sample - this step's
 * senders into `senders_scratch` (device int32 [nb_envs][nb_houses][nb_comm], the draws of mdr_env_comm_draws; may be NULL otherwise);
 * the actor kernel then stages one lane per agent and GATHERS its nb_comm records through the table.  Same draws, outputs and
 * `rows_out` as mdr_env_actor_sample (rows bit for bit those of mdr_env_obs_vector); the optional MESSAGE columns stay unsupported
 * (-4: with 10 senders they do not fit the 64 features of a staged row).  Circular neighbours are accepted too (NULL links). */
int mdr_env_actor_sample_links(mdr_env_t *env, const mdr_obs_spec_t *spec, const mdr_actor_t *actor, float *msg_scratch,
                               int32_t *senders_scratch, uint64_t seed, uint64_t step, const int32_t *step_dev, uint8_t *action,
                               float *a_prob, float *probs, float *rows_out, void *stream);

/* The Monte-Carlo return scan of PPO.update (agents/ppo.py:123-134) for every agent at once: backwards over t,
 * R <- reward[t] + gamma * (done[t] ? bootstrap[t] : R).  `reward`, `out` float [nb_steps][nb_agents]; `done` uint8 of that
 * shape or NULL (no restarts); `bootstrap` float of that shape (the critic's value of the next state where done) or NULL
 * (restart from 0: zero_eoepisode_return). */
int mdr_discounted_returns(const float *reward, const uint8_t *done, const float *bootstrap, float gamma, int32_t nb_steps,
                           int64_t nb_agents, float *out, void *stream);

#ifdef __cplusplus
}
#endif
#endif
