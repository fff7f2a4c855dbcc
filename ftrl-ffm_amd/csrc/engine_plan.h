// engine_plan.h -- the field-pair partition of a sharded model (ffm_engine_shard_plan): host
// arithmetic only.  Part of engine.hip's translation unit (included inside its anonymous namespace).

// ---- field-pair partition (include/ffm_engine.h: ffm_engine_shard_plan) -----------------------
// The fields are cut into g contiguous groups; the unit of ownership is a BLOCK (a, b), a <= b: all
// field pairs with one field in group a and the other in group b.  Every shard gets a set of blocks
// such that, for each group, the partner groups it owns form one interval -- so every field's owned
// partner fields are ONE contiguous range and the owned slots of a record are contiguous.
//   1 shard : everything.
//   2 shards: g = 2; {(0,1)} | {(0,0),(1,1)}.
//   4 shards: g = 4; {(0,1),(2,3)} | {(0,2),(1,3)} | {(0,3),(1,2)} | the four diagonal blocks.
//   8 shards: g = 4; the six off-diagonal blocks one each | {(0,0),(1,1)} | {(2,2),(3,3)}: every
//             shard needs the columns of only two groups (about half of the fields); 100 vs 92.6
//             cross-field pairs on the busiest shard at 39 fields.
//   other   : g = n "strips": shard r owns the blocks (r, s), s >= r, with the group sizes chosen
//             so that the shards' pair counts are as even as the field count allows.
struct ShardPlan {
  int n_fields = 0, n_shards = 1;
  std::vector<int> own_lo, own_n;  // [shard][field]
  std::vector<int> lin_owner;      // [field] shard that owns the field's linear terms
  int bias_owner = 0;
  std::vector<long long> pairs;    // [shard] cross-field pairs owned (load measure)
  int lo(int r, int f) const { return own_lo[static_cast<size_t>(r) * n_fields + f]; }
  int n(int r, int f) const { return own_n[static_cast<size_t>(r) * n_fields + f]; }
  bool owns(int r, int fa, int fb) const { return static_cast<unsigned>(fb - lo(r, fa)) < static_cast<unsigned>(n(r, fa)); }
};

static ShardPlan make_shard_plan(int F, int N, bool field_map) {
  ShardPlan p;
  p.n_fields = F;
  p.n_shards = N;
  p.own_lo.assign(static_cast<size_t>(N) * F, 0);
  p.own_n.assign(static_cast<size_t>(N) * F, 0);
  p.lin_owner.assign(F, 0);
  p.pairs.assign(N, 0);
  std::vector<int> gb;                                  // group boundaries, g + 1 entries
  std::vector<std::vector<std::pair<int, int>>> blocks(N);
  auto even_groups = [&](int g) { gb.clear(); for (int j = 0; j <= g; j++) gb.push_back(static_cast<int>(static_cast<long long>(j) * F / g)); };
  if (N == 1) {
    even_groups(1);
    blocks[0] = {{0, 0}};
  } else if (N == 2 && F >= 2) {
    even_groups(2);
    blocks[0] = {{0, 1}};
    blocks[1] = {{0, 0}, {1, 1}};
  } else if (N == 4 && F >= 4) {
    even_groups(4);
    blocks[0] = {{0, 1}, {2, 3}};
    blocks[1] = {{0, 2}, {1, 3}};
    blocks[2] = {{0, 3}, {1, 2}};
    blocks[3] = {{0, 0}, {1, 1}, {2, 2}, {3, 3}};
  } else if (N == 8 && F >= 4) {
    even_groups(4);
    blocks[0] = {{0, 1}}; blocks[1] = {{0, 2}}; blocks[2] = {{0, 3}};
    blocks[3] = {{1, 2}}; blocks[4] = {{1, 3}}; blocks[5] = {{2, 3}};
    blocks[6] = {{0, 0}, {1, 1}};
    blocks[7] = {{2, 2}, {3, 3}};
  } else {
    // strips: choose the boundaries greedily so that shard r's cross-field pair count
    // |G_r| * (F - end_r) + C(|G_r|, 2) tracks what is left divided by the shards left
    gb.assign(1, 0);
    long long left = static_cast<long long>(F) * (F - 1) / 2;
    for (int r = 0; r < N; r++) {
      const int b0 = gb.back();
      int b1 = b0;
      if (r == N - 1) {
        b1 = F;
      } else {
        const double target = static_cast<double>(left) / (N - r);
        long long best_cnt = 0;
        for (int c = b0; c <= F; c++) {
          const long long sz = c - b0, cnt = sz * (F - c) + sz * (sz - 1) / 2;
          if (c == b0 || std::abs(static_cast<double>(cnt) - target) <= std::abs(static_cast<double>(best_cnt) - target)) { b1 = c; best_cnt = cnt; }
          if (static_cast<double>(cnt) > target) break;
        }
        left -= best_cnt;
      }
      gb.push_back(b1);
      for (int s2 = r; s2 < N; s2++) blocks[r].push_back({r, s2});
    }
  }
  const int g = static_cast<int>(gb.size()) - 1;
  for (int r = 0; r < N; r++) {
    std::vector<int> pmin(g, g), pmax(g, -1);  // partner-group interval of every group on shard r
    for (auto [a, b] : blocks[r]) {
      if (b >= g || a >= g) continue;
      pmin[a] = std::min(pmin[a], b); pmax[a] = std::max(pmax[a], b);
      pmin[b] = std::min(pmin[b], a); pmax[b] = std::max(pmax[b], a);
      const long long sa = gb[a + 1] - gb[a], sb = gb[b + 1] - gb[b];
      p.pairs[r] += a == b ? sa * (sa - 1) / 2 : sa * sb;
    }
    for (int a = 0; a < g; a++) {
      if (pmax[a] < 0) continue;
      for (int f = gb[a]; f < gb[a + 1]; f++) {
        p.own_lo[static_cast<size_t>(r) * F + f] = gb[pmin[a]];
        p.own_n[static_cast<size_t>(r) * F + f] = gb[pmax[a] + 1] - gb[pmin[a]];
      }
    }
  }
  // Without a field map the bias and all linear terms go to the least loaded shard (a shard then
  // cannot tell a feature's field from its id).  With one, every field's linear terms go to the
  // least loaded shard that keeps that column, and the bias to the least loaded shard after that;
  // loads in units of one field pair.  Re-fitted in round 6 to the emulated per-rank steps of round 5
  // (39 fields / 8 shards, 65 536-row blocks; step = 1.22 ms + 5 us per unit): off-diagonal shards
  // 90 / 100 pairs -> 1.67 / 1.72 ms whatever linear terms they carry (a field's linear terms ~ 0.3
  // pairs; the bias, a two-level fold since round 5, ~ 1 -- rounds 2-4 measured 2 and 15), the two
  // shards made of DIAGONAL blocks -- pairs inside a group: 81 / 90 of them -> 1.76 / 1.81 ms -- cost
  // 1.30 per pair (round 3's estimate: 1.12).
  std::vector<double> load(p.pairs.begin(), p.pairs.end());
  for (int r = 0; r < N; r++) {
    bool diagonal = false;
    for (auto [a, b] : blocks[r]) diagonal = diagonal || (a == b && a < g);
    if (diagonal && field_map) load[r] += 0.30 * static_cast<double>(p.pairs[r]);
  }
  p.bias_owner = static_cast<int>(std::min_element(load.begin(), load.end()) - load.begin());
  if (field_map) {
    for (int f = 0; f < F; f++) {
      int best = -1;
      for (int r = 0; r < N; r++)
        if (p.n(r, f) > 0 && (best < 0 || load[r] < load[best])) best = r;
      if (best < 0) best = p.bias_owner;
      p.lin_owner[f] = best;
      load[best] += 0.3;
    }
    p.bias_owner = static_cast<int>(std::min_element(load.begin(), load.end()) - load.begin());
  } else {
    for (int f = 0; f < F; f++) p.lin_owner[f] = p.bias_owner;
  }
  return p;
}
