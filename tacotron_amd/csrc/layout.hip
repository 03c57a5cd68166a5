// layout.hip -- host-only: parameter table (TF variable order), transposed-weight table, workspace table.
#include "layout.h"
#include "kernels.h"

#include <stdarg.h>
#include <string.h>

static thread_local char g_err[512] = "ok";
void taco_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* taco_last_error_string(void) { return g_err; }

// ---- tail events (common.h) ----
namespace {
constexpr int kTailRing = 64, kTailStreams = 8, kPlanWords = 8, kPlans = 32;   // (a plan covers 64 * kPlanWords launches per stream)
hipStream_t const kDepMany = reinterpret_cast<hipStream_t>(~(uintptr_t)0);
struct TailTrack {
  bool used = false;
  hipStream_t s = nullptr;
  hipEvent_t ring[kTailRing] = {};
  int next = 0;
  int tail_slot = -1;          // ring slot that still owns `tail` (-1: stolen, not a ring event, or no tail)
  hipEvent_t tail = nullptr;   // rides on the last launch on s; nullptr once anything it does not cover was enqueued behind it
  bool has_dep = false;        // behind the last launch s was made to wait for events of stream `dep` (kDepMany: of several streams):
  hipStream_t dep = nullptr;   //   `tail` then still covers everything a fork TO `dep` has to wait for (dep's own order covers the rest)
  int launches = 0;            // launches on s in this scope
  int tail_idx = -1;           // launch index of `tail`
  bool declined = false;       // the last launch on s carried no event because the plan did not ask for one
  int dev = 0;                 // device the ring's events belong to
  uint64_t stamp = 0;          // last use (least-recently-used take-over when a caller keeps handing in new streams)
  bool armed = false;          // profiling bracket: the next launch carries these two events
  hipEvent_t arm_start = nullptr, arm_stop = nullptr;
  int arm_launches = 0;
};
// Which launches of a call need an event is learned, not declared: the first call of a kind / shape puts an event on EVERY launch
// and notes the (stream, launch index) pairs a fork, join or segment actually consumed; later calls put events on those only
// (an event on all ~90 launches of a step costs what the ~17 markers it replaces cost: measured, profiles/r06_tail_events_ab.txt).
// A fork that finds no event because the plan declined it falls back to a recorded marker -- always correct -- and the plan is
// learned again by the next call.
struct TailPlan {
  bool used = false, learned = false;
  uint64_t key = 0;
  uint64_t bits[kTailStreams][kPlanWords] = {};
};
thread_local TailTrack g_tail[kTailStreams];
thread_local TailPlan g_plans[kPlans];
thread_local TailPlan* g_plan = nullptr;
thread_local int g_plan_next = 0;
thread_local bool g_tail_on = false, g_learning = false;
thread_local uint64_t g_tail_clock = 0;
void tail_drop(TailTrack* t);
TailTrack* tail_find(hipStream_t s, bool create) {
  for (TailTrack& t : g_tail)
    if (t.used && t.s == s) {
      t.stamp = ++g_tail_clock;
      return &t;
    }
  if (!create) return nullptr;
  int dev = 0;
  (void)hipGetDevice(&dev);
  TailTrack* lru = nullptr;
  for (TailTrack& t : g_tail) {
    if (!t.used) {
      t.used = true;
      t.s = s;
      t.dev = dev;
      t.stamp = ++g_tail_clock;
      return &t;
    }
    // a caller that hands in ever new streams: the least recently used entry of the SAME device is taken over (its ring events
    // are not bound to a stream); entries of streams that launched in this scope are left alone
    if (t.dev == dev && t.launches == 0 && !t.tail && !t.armed && (!lru || t.stamp < lru->stamp)) lru = &t;
  }
  if (lru) {
    lru->s = s;
    lru->stamp = ++g_tail_clock;
    tail_drop(lru);
    lru->tail_idx = -1;
    lru->declined = false;
  }
  return lru;
}
void tail_consumed(TailTrack* t) {
  if (g_learning && g_plan && t->tail_idx >= 0 && t->tail_idx < 64 * kPlanWords)
    g_plan->bits[t - g_tail][t->tail_idx >> 6] |= 1ull << (t->tail_idx & 63);
}
void tail_drop(TailTrack* t) {
  t->tail = nullptr;
  t->tail_slot = -1;
  t->has_dep = false;
  t->dep = nullptr;
}
}  // namespace

hipEvent_t taco_tail_take(hipStream_t s, hipEvent_t* start) {
  *start = nullptr;
  if (!g_tail_on) return nullptr;
  TailTrack* t = tail_find(s, true);
  if (!t) return nullptr;   // (more streams than the table holds: those launch plainly and fork through recorded events)
  const int idx = t->launches++;
  ++t->arm_launches;
  t->has_dep = false;       // (this launch is ordered behind every wait enqueued so far: its event covers them)
  t->dep = nullptr;
  if (t->armed) {           // a profiling bracket's pair: timing events the ring does not own
    t->armed = false;
    *start = t->arm_start;
    t->tail = t->arm_stop;
    t->tail_slot = -1;
    t->tail_idx = idx;
    t->declined = false;
    return t->tail;
  }
  const bool want = g_learning || (g_plan && idx < 64 * kPlanWords && ((g_plan->bits[t - g_tail][idx >> 6] >> (idx & 63)) & 1));
  if (!want) {
    tail_drop(t);
    t->declined = true;
    return nullptr;
  }
  const int slot = t->next;
  t->next = (slot + 1) % kTailRing;
  t->declined = false;
  if (!t->ring[slot] && hipEventCreateWithFlags(&t->ring[slot], hipEventDisableTiming) != hipSuccess) {
    t->ring[slot] = nullptr;
    tail_drop(t);
    return nullptr;
  }
  t->tail = t->ring[slot];
  t->tail_slot = slot;
  t->tail_idx = idx;
  return t->tail;
}
hipEvent_t taco_tail_event(hipStream_t s, hipStream_t for_stream) {
  if (!g_tail_on) return nullptr;
  TailTrack* t = tail_find(s, false);
  if (!t) return nullptr;
  if (t->tail && t->has_dep && (t->dep == kDepMany || t->dep != for_stream)) return nullptr;   // (waits behind the launch that `for_stream` does not inherit by its own order)
  if (t->tail) tail_consumed(t);
  else if (t->declined && g_plan && !g_learning) g_plan->learned = false;   // (mispredicted: this call falls back, the next one learns)
  return t->tail;
}
hipEvent_t taco_tail_steal(hipStream_t s, hipEvent_t give, bool* owned) {
  *owned = false;
  hipEvent_t e = taco_tail_event(s, nullptr);
  if (!e) return nullptr;
  TailTrack* t = tail_find(s, false);
  if (t->tail_slot >= 0 && t->ring[t->tail_slot] == e) {
    t->ring[t->tail_slot] = give;   // (an event lives in exactly one place: a ring slot or its new owner)
    t->tail_slot = -1;
    *owned = true;
  }
  return e;   // (not owned: a profiling bracket's stop event, or one that was stolen before -- whoever waits for it does so before it is bound again)
}
void taco_tail_touch(hipStream_t s) {
  TailTrack* t = tail_find(s, false);
  if (t) {
    tail_drop(t);
    t->declined = false;
  }
}
void taco_tail_open(uint64_t key) {
  const char* e = getenv("TACO_TAIL_EVENTS");
  g_tail_on = !(e && atoi(e) == 0);
  g_plan = nullptr;
  g_learning = false;
  for (TailTrack& t : g_tail) {
    tail_drop(&t);
    t.launches = 0;
    t.tail_idx = -1;
    t.declined = false;
    t.armed = false;
  }
  if (!g_tail_on) return;
  for (TailPlan& p : g_plans)
    if (p.used && p.key == key) g_plan = &p;
  if (!g_plan) {
    g_plan = &g_plans[g_plan_next];
    g_plan_next = (g_plan_next + 1) % kPlans;
    *g_plan = TailPlan();
    g_plan->used = true;
    g_plan->key = key;
  }
  if (!g_plan->learned || (e && atoi(e) == 2)) {   // (TACO_TAIL_EVENTS=2: an event on every launch, always)
    g_learning = true;
    for (auto& row : g_plan->bits)
      for (uint64_t& w : row) w = 0;
  }
}
void taco_tail_close() {
  if (g_tail_on && g_learning && g_plan) g_plan->learned = true;
  g_tail_on = false;
  g_learning = false;
  g_plan = nullptr;
  for (TailTrack& t : g_tail) {
    tail_drop(&t);
    t.armed = false;
  }
}
bool taco_tail_wait(hipStream_t waiter, hipStream_t producer) {
  hipEvent_t e = taco_tail_event(producer, waiter);
  if (!e) return false;
  if (hipStreamWaitEvent(waiter, e, 0) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  // the wait itself is not covered by the waiter's last launch -- except for a fork back TO the producer, whose own order covers it
  if (TailTrack* w = tail_find(waiter, false)) {
    if (!w->has_dep) {
      w->has_dep = true;
      w->dep = producer;
    } else if (w->dep != producer) {
      w->dep = kDepMany;
    }
  }
  return true;
}
bool taco_tail_arm_timing(hipStream_t s, hipEvent_t start, hipEvent_t stop) {
  if (!g_tail_on) return false;
  TailTrack* t = tail_find(s, true);
  if (!t) return false;
  t->armed = true;
  t->arm_start = start;
  t->arm_stop = stop;
  t->arm_launches = 0;
  return true;
}
int taco_tail_disarm_timing(hipStream_t s) {
  TailTrack* t = tail_find(s, false);
  if (!t) return 0;
  t->armed = false;
  return t->arm_launches;
}

extern "C" int taco_version(void) { return TACO_VERSION; }

int validate_shape(const TacoShape* s) {
  TACO_REQUIRE(s != nullptr, "shape is null");
  TACO_REQUIRE(s->B >= 1 && s->B <= 4096, "B=%d out of range", s->B);
  TACO_REQUIRE(s->Tt >= 1 && s->Tt <= 4096, "Tt=%d out of range", s->Tt);
  TACO_REQUIRE(s->Td >= 1 && s->Td <= 8192, "Td=%d out of range", s->Td);
  TACO_REQUIRE(s->r >= 1 && s->r <= 5, "r=%d out of range (1..5)", s->r);
  TACO_REQUIRE(s->V >= 1, "V=%d out of range", s->V);
  TACO_REQUIRE(s->S >= 0 && s->S <= 100000, "S=%d out of range", s->S);
  return TACO_OK;
}

namespace {
struct Adder {
  std::vector<TacoTensorInfo>* rows;
  int64_t off = 0;
  int64_t align = 1;
  int64_t add(const std::string& name, std::initializer_list<int64_t> dims) {
    TacoTensorInfo r;
    memset(&r, 0, sizeof(r));
    snprintf(r.name, sizeof(r.name), "%s", name.c_str());
    int64_t sz = 1;
    int i = 0;
    for (int64_t d : dims) {
      r.dims[i++] = (int32_t)d;
      sz *= d;
    }
    r.ndim = i;
    off = (off + align - 1) / align * align;
    r.offset = off;
    r.size = sz;
    if (rows) rows->push_back(r);
    off += sz;
    return r.offset;
  }
};

void add_dense(Adder& a, const std::string& n, int in, int out, bool bias, DenseP& d) {
  d.in = in;
  d.out = out;
  d.w = a.add(n + "/kernel", {in, out});
  d.b = bias ? a.add(n + "/bias", {out}) : -1;
}
void add_gru(Adder& a, const std::string& n, int cin, int h, GruP& g) {
  g.cin = cin;
  g.h = h;
  g.wg = a.add(n + "/gates/kernel", {cin + h, 2 * h});
  g.bg = a.add(n + "/gates/bias", {2 * h});
  g.wc = a.add(n + "/candidate/kernel", {cin + h, h});
  g.bc = a.add(n + "/candidate/bias", {h});
}
void add_cbhg(Adder& a, const std::string& p, int K, int cin, int c1, int c2, bool spk, CbhgP& c) {
  c.spk = spk;
  c.K = K;
  c.cin = cin;
  c.c1 = c1;
  c.c2 = c2;
  for (int k = 1; k <= K; ++k) {
    c.bank_w[k - 1] = a.add(p + "bank_" + std::to_string(k) + "/kernel", {k, cin, kCb});
    c.bank_b[k - 1] = a.add(p + "bank_" + std::to_string(k) + "/bias", {kCb});
  }
  c.bank_g = a.add(p + "bank_bn/gamma", {K * kCb});
  c.bank_be = a.add(p + "bank_bn/beta", {K * kCb});
  c.p1_w = a.add(p + "proj1/kernel", {3, K * kCb, c1});
  c.p1_b = a.add(p + "proj1/bias", {c1});
  c.p1_g = a.add(p + "proj1_bn/gamma", {c1});
  c.p1_be = a.add(p + "proj1_bn/beta", {c1});
  c.p2_w = a.add(p + "proj2/kernel", {3, c1, c2});
  c.p2_b = a.add(p + "proj2/bias", {c2});
  c.p2_g = a.add(p + "proj2_bn/gamma", {c2});
  c.p2_be = a.add(p + "proj2_bn/beta", {c2});
  for (int l = 0; l < 4; ++l) {
    const std::string hp = p + "highway_" + std::to_string(l) + "/";
    c.has_adapt[l] = spk || (l == 0 && c2 != kCb);
    if (spk) {
      add_dense(a, hp + "spk", 16, kCb, true, c.spkd[l]);
      add_dense(a, hp + "adapt", 2 * kCb, kCb, true, c.adapt[l]);
    } else if (c.has_adapt[l]) {
      add_dense(a, hp + "adapt", c2, kCb, true, c.adapt[l]);
    }
    add_dense(a, hp + "T", kCb, kCb, true, c.hwT[l]);
    add_dense(a, hp + "H", kCb, kCb, true, c.hwH[l]);
  }
  if (spk) add_dense(a, p + "gru_init", 16, kCb, true, c.gru_init);
  add_gru(a, p + "bigru/fw", kCb, kCb, c.fw);
  add_gru(a, p + "bigru/bw", kCb, kCb, c.bw);
}
}  // namespace

void build_param_layout(const TacoShape& s, ParamLayout& L) {
  L.rows.clear();
  Adder a{&L.rows};
  const int R80 = kMel * s.r;
  const bool multi = s.S > 1;
  L.emb = a.add("embedding", {s.V, kEmbed});
  L.spk_embed = multi ? a.add("speaker_embed", {s.S, 16}) : -1;
  add_dense(a, "encoder/pre_net/dense", kEmbed, kPre1, true, L.enc_pre1);
  add_dense(a, "encoder/pre_net/dense_1", kPre1, kPre2, true, L.enc_pre2);
  add_cbhg(a, "encoder/cbhg/", 16, kPre2, kCb, kCb, multi, L.enc);
  L.mem_w = a.add("decoder/memory_layer/kernel", {2 * kCb, kAtt});
  add_dense(a, "decoder/pre_net/dense", kMel, kPre1, true, L.dec_pre1);
  add_dense(a, "decoder/pre_net/dense_1", kPre1, kPre2, true, L.dec_pre2);
  add_dense(a, "decoder/in_proj", kPre2 + kAtt, kDec, true, L.in_proj);
  for (int l = 0; l < 3; ++l) add_gru(a, "decoder/gru_" + std::to_string(l), kDec, kDec, L.gru[l]);
  add_dense(a, "decoder/out_proj", kDec, R80, true, L.out_proj);
  L.q_w = a.add("decoder/query_layer/kernel", {R80, kAtt});
  L.att_v = a.add("decoder/attention_v", {kAtt});
  L.att_w = a.add("decoder/attention_layer/kernel", {R80 + kAtt, kAtt});
  add_cbhg(a, "post/cbhg/", 8, kMel, 256, kMel, false, L.post);
  add_dense(a, "post/dense", 2 * kCb, kFft, true, L.post_dense);
  L.total = a.off;
}

static void trans_cbhg(Adder& a, const CbhgP& c, CbhgT& t) {
  for (int k = 1; k <= c.K; ++k) t.bank[k - 1] = a.add("", {k, kCb, c.cin});
  t.p1 = a.add("", {3, c.c1, c.K * kCb});
  t.p2 = a.add("", {3, c.c2, c.c1});
  for (int l = 0; l < 4; ++l) {
    t.adapt[l] = c.has_adapt[l] ? a.add("", {kCb, c.spk ? kCb : c.c2}) : -1;
    t.adapt_s[l] = c.spk ? a.add("", {kCb, kCb}) : -1;
    t.spkd[l] = c.spk ? a.add("", {kCb, 16}) : -1;
  }
  t.gru_init = c.spk ? a.add("", {kCb, 16}) : -1;
  for (int l = 0; l < 4; ++l) t.hw[l] = a.add("", {2 * kCb, kCb});
  t.gru_x = a.add("", {6 * kCb, kCb});
  for (int d = 0; d < 2; ++d) {
    t.wghT[d] = a.add("", {2 * kCb, kCb});
    t.wchT[d] = a.add("", {kCb, kCb});
  }
}

void build_trans_layout(const TacoShape& s, const ParamLayout& P, TransLayout& T) {
  Adder a{nullptr};
  a.align = 4;
  const int R80 = kMel * s.r;
  T.enc_pre1 = a.add("", {kPre1, kEmbed});
  T.enc_pre2 = a.add("", {kPre2, kPre1});
  trans_cbhg(a, P.enc, T.enc);
  T.mem_w = a.add("", {kAtt, 2 * kCb});
  T.dec_pre1 = a.add("", {kPre1, kMel});
  T.dec_pre2 = a.add("", {kPre2, kPre1});
  T.in_proj = a.add("", {kDec, kPre2 + kAtt});
  for (int l = 0; l < 3; ++l) {
    T.gw[l] = a.add("", {2 * kDec, 2 * kDec});
    T.cw[l] = a.add("", {kDec, 2 * kDec});
  }
  T.out_proj = a.add("", {R80, kDec});
  T.q_w = a.add("", {kAtt, R80});
  T.att_w = a.add("", {kAtt, R80 + kAtt});
  trans_cbhg(a, P.post, T.post);
  T.post_dense = a.add("", {1028, 2 * kCb});   // 1025 rows + 3 zero rows: the backward GEMM runs with K padded to 1028
  T.total = a.off;
}

static void ws_cbhg(Adder& a, const std::string& p, const CbhgP& c, int64_t M, int64_t B, bool train, CbhgWs& w) {
  (void)train;
  w.bank = a.add(p + "bank", {M, c.K * kCb});
  w.pool = a.add(p + "pool", {M, c.K * kCb});
  w.pj1pre = a.add(p + "pj1pre", {M, c.c1});
  w.pj1 = a.add(p + "pj1", {M, c.c1});
  w.pj2pre = a.add(p + "pj2pre", {M, c.c2});
  w.res = a.add(p + "res", {M, c.c2});
  w.h[0] = w.res;   // h[l] = input of highway layer l (before the optional adapter); hx[l] = after it
  for (int l = 1; l <= 4; ++l) w.h[l] = a.add(p + "h" + std::to_string(l), {M, kCb});
  for (int l = 0; l < 4; ++l) w.hx[l] = c.has_adapt[l] ? a.add(p + "hx" + std::to_string(l), {M, kCb}) : w.h[l];
  for (int l = 0; l < 4; ++l) w.th[l] = a.add(p + "th" + std::to_string(l), {M, 2 * kCb});
  w.xg = a.add(p + "xg", {M, 6 * kCb});
  w.out = a.add(p + "out", {M, 2 * kCb});
  w.ruc = a.add(p + "ruc", {M, 6 * kCb});
  {
    // speaker sites (ops.py:101-115): sv[0..3] | h0 as ONE (5,B,128) block, likewise their gradients (dsm2) -- the ReLU backward of all
    // five is one launch; rowb / dsm per layer
    const int64_t blk = B * kCb;
    const int64_t svh = c.spk ? a.add(p + "sv_h0", {5, B, kCb}) : -1;
    const int64_t rb = c.spk ? a.add(p + "rowb", {4, B, kCb}) : -1;
    for (int l = 0; l < 4; ++l) {
      w.sv[l] = c.spk ? svh + l * blk : -1;
      w.rowb[l] = c.spk ? rb + l * blk : -1;
    }
    w.h0 = c.spk ? svh + 4 * blk : -1;
    w.dh0 = c.spk ? a.add(p + "dh0", {2, B, kCb}) : -1;
    w.dsmall = c.spk ? a.add(p + "dsmall", {B, kCb}) : -1;
    w.dsmall2 = c.spk ? a.add(p + "dsmall2", {B, kCb}) : -1;
    const int64_t d1 = c.spk ? a.add(p + "dsm", {4, B, kCb}) : -1;
    const int64_t d2 = c.spk ? a.add(p + "dsm2", {5, B, kCb}) : -1;
    for (int l = 0; l < 4; ++l) w.dsm[l] = c.spk ? d1 + l * blk : -1;
    for (int l = 0; l < 5; ++l) w.dsm2[l] = c.spk ? d2 + l * blk : -1;
    w.dspk_part = c.spk ? a.add(p + "dspk_part", {5, B, 16}) : -1;
  }
}

void build_ws_layout(const TacoShape& s, bool train, const TransLayout& T, WsLayout& W) {
  W.rows.clear();
  Adder a{&W.rows};
  a.align = 64;  // 256-byte alignment of every workspace tensor
  ParamLayout P;
  build_param_layout(s, P);
  const int R80 = kMel * s.r;
  const int64_t M1 = (int64_t)s.B * s.Tt, MD = (int64_t)s.B * s.Td, M2 = MD * s.r;
  W.emb = a.add("enc.emb", {M1, kEmbed});
  W.p1 = a.add("enc.p1", {M1, kPre1});
  W.p2 = a.add("enc.p2", {M1, kPre2});
  W.spk_e = s.S > 1 ? a.add("enc.spk_e", {s.B, 16}) : -1;
  W.dspk_e = s.S > 1 ? a.add("enc.dspk_e", {s.B, 16}) : -1;
  ws_cbhg(a, "enc.", P.enc, M1, s.B, train, W.enc);
  W.values = a.add("dec.values", {M1, kAtt});
  W.keys = a.add("dec.keys", {M1, kAtt});
  W.vwx = a.add("dec.vwx", {M1, kDec});        // values . Wx_c: the context's path into the input projection, per memory row
  W.vwg = a.add("dec.vwg", {M1, 2 * kDec});    // values . Wx_c Wg0_x: ... and into GRU-1's gates
  W.vcen = train ? a.add("dec.vcen", {M1, kAtt}) : -1;    // values minus their per-sequence mean row
  W.vwxc = train ? a.add("dec.vwxc", {M1, kDec}) : -1;    // vcen . Wx_c: what the backward kernel scores d alignments against
  W.stash = train ? a.add("dec.stash", {MD, kStRec}) : -1;
  W.prein = train ? a.add("dec.prein", {MD, kMel}) : -1;
  W.xchg = a.add("dec.xchg", {decoder_xchg_bytes(s.B, s.Tt) / 4});
  W.err = a.add("dec.err", {512});   // [0],[1] error words; floats 16.. = optional phase trace
  {
    const int KX = kPre2 + R80 + kAtt, NO = dec_out_cols(s.r);
    W.dc_wx = a.add("dec.comp.wx", {KX, kDec});                 // [Wi_p ; Wa Wi_a]: x = [p2 ; out ; ctx] Wx + bi
    W.dc_wg0 = a.add("dec.comp.wg0", {KX + kDec, 2 * kDec});    // [Wx_po Wg0_x ; Wg0_h ; Wx_c Wg0_x]  (rows: p2, out | h1 | ctx)
    W.dc_bg0 = a.add("dec.comp.bg0", {2 * kDec});               // bg0 + bi Wg0_x
    W.dc_wo = a.add("dec.comp.wo", {kDec, NO});                 // [Wo Wq | Wo | 0]
    W.dc_bo = a.add("dec.comp.bo", {NO});                       // [bo Wq | bo | 0]
    W.dc_wp1o = a.add("dec.comp.wp1o", {kDec, kPre1});          // Wo[:, last frame] W1
    W.dc_bp1o = a.add("dec.comp.bp1o", {kPre1});                // bo[last frame] W1 + b1
  }
  ws_cbhg(a, "post.", P.post, M2, s.B, train, W.post);
  W.wd_pad = a.add("post.wd_pad", {2 * kCb, 1028});   // post/dense kernel re-pitched to a 16-byte-aligned leading dimension
  {
    const int64_t e1 = M1 * (int64_t)P.enc.c1, e2 = M2 * (int64_t)P.post.c1;
    W.tapsplit = a.add("tapsplit", {4, e1 > e2 ? e1 : e2});
    W.tapsplit_floats = 4 * (e1 > e2 ? e1 : e2);
  }
  W.loss = a.add("loss", {4 + 2 * kLossParts});   // [0..3] total / seq2seq / output; then the two terms' per-block partials
  if (train) {
    const int64_t Mx = M1 > M2 ? M1 : M2;
    W.ds2s = a.add("bwd.ds2s", {MD, R80});
    W.dout_pad = a.add("bwd.dout_pad", {M2, 1028});
    W.paramsT = a.add("bwd.paramsT", {T.total});
    W.gstash = a.add("bwd.gstash", {MD, kGsRec});
    // one (M1, 512) buffer: columns [0,256) d keys (decoder backward kernel), [256,512) E[b] = sum_t al_{t-1}^T dx_t; the
    // encoder-output gradient is then ONE product [d keys | E] . [Wm^T ; Wx_c^T]
    W.dkeys = a.add("bwd.dkeys_e", {M1, 2 * kAtt});
    W.dvalues = W.dkeys + kAtt;
    W.ds2s_tot = a.add("bwd.ds2s_tot", {MD, R80});
    W.bc_fa = a.add("bwd.comp.fa", {kDec, dec_fan_cols(s.r)});      // [Wx_o^T | 0]
    W.bc_wmx = a.add("bwd.comp.wmx", {2 * kAtt, 2 * kCb});          // [Wm^T ; Wx_c^T]
    W.bc_wxct = W.bc_wmx + (int64_t)kAtt * 2 * kCb;
    W.bc_wdx = a.add("bwd.comp.wdx", {kDec, kDec});                 // Wx_o^T Wo^T: dx_{t+1} -> d(x + h3)_t
    W.bc_wot = a.add("bwd.comp.wot", {R80 + 2 * kAtt, kDec});       // [Wo^T ; (Wo Wq)^T ; (Wo_f W1)^T]
    W.bc_g = a.add("bwd.comp.g", {R80 + kAtt, kDec});               // sum_t [out_t ; ctx_t]^T dx_{t+1}
    W.bc_h1 = a.add("bwd.comp.h1", {kDec, kAtt});                   // sum_t (x + h3)_t^T dq_t
    W.bc_h2 = a.add("bwd.comp.h2", {kDec, kPre1});                  // sum_t (x + h3)_t^T dp1s_t
    W.bc_cq = a.add("bwd.comp.cq", {kAtt});                         // sum_t dq_t
    W.bc_cp = a.add("bwd.comp.cp", {kPre1});                        // sum_t dp1s_t
    W.dattv = a.add("bwd.dattv_rows", {s.B, kAtt});                 // per-row d attention_v, summed in row order
    W.post_dpj1 = a.add("bwd.post.dpj1", {M2, P.post.c1});
    W.post_dz1 = a.add("bwd.post.dz1", {M2, P.post.c1});
    W.post_dpool = a.add("bwd.post.dpool", {M2, P.post.K * kCb});
    W.post_dx = a.add("bwd.post.dx", {M2, kMel});
    // encoder backward operands that outlive their stage: its weight-gradient GEMMs run on the side stream, beside the activation-
    // gradient chain (model.hip), so the chain may not reuse their operands in place
    W.enc_dpj1 = a.add("bwd.enc.dpj1", {M1, P.enc.c1});
    W.enc_dz1 = a.add("bwd.enc.dz1", {M1, P.enc.c1});
    W.enc_dpool = a.add("bwd.enc.dpool", {M1, P.enc.K * kCb});
    W.enc_dx = a.add("bwd.enc.dx", {M1, kCb});
    W.pre_dz2 = a.add("bwd.pre.dz2", {M1, kPre2});
    W.pre_dz1 = a.add("bwd.pre.dz1", {M1, kPre1});
    W.pre_demb = a.add("bwd.pre.demb", {M1, kEmbed});
    W.gA = a.add("bwd.gA", {Mx, 16 * kCb});
    W.gB = a.add("bwd.gB", {Mx, 16 * kCb});
    W.gC = a.add("bwd.gC", {Mx, 6 * kCb});
    W.gD = a.add("bwd.gD", {Mx, 2 * kCb});
    W.gE = a.add("bwd.gE", {Mx, 2 * kCb});
    W.gF = a.add("bwd.gF", {Mx, 2 * kCb});
    W.gG = a.add("bwd.gG", {Mx, 2 * kCb});
    W.scratch = a.add("bwd.scratch", {64});
  } else {
    W.ds2s = W.dout_pad = W.paramsT = W.gstash = W.dkeys = W.dvalues = W.ds2s_tot = -1;
    W.bc_wxct = W.bc_wdx = W.bc_wmx = -1;
    W.bc_fa = W.bc_wot = W.bc_g = W.bc_h1 = W.bc_h2 = W.bc_cq = W.bc_cp = W.dattv = -1;
    W.post_dpj1 = W.post_dz1 = W.post_dpool = W.post_dx = -1;
    W.enc_dpj1 = W.enc_dz1 = W.enc_dpool = W.enc_dx = W.pre_dz2 = W.pre_dz1 = W.pre_demb = -1;
    W.gA = W.gB = W.gC = W.gD = W.gE = W.gF = W.gG = W.scratch = -1;
  }
  {
    int64_t fl = 0;
    for_each_weight_image(P, T, train, [&](int, int64_t, int, int, int64_t, int, int taps, int K, int N, bool) { fl += weight_image_floats(taps, K, N); });
    W.wimg = a.add("gemm.wimg", {fl});
  }
  W.total = (a.off + 63) / 64 * 64;
}
