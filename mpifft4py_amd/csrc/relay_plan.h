// relay_plan.h -- how one message of a SUB-GROUP exchange is striped over the links of a fully connected node
// (host-only arithmetic, shared by the IPC transport and the device-free query mfft_plan_relay_schedule).
//
// The pencil decompositions exchange inside groups of g = P1 or P2 ranks (pencil.py:741-750, 1324-1333: Alltoallw on
// comm0 / comm1), all groups at the same time.  On MI355X every pair of GPUs has its own xGMI link, so a rank that
// talks to g - 1 group peers leaves its P - g links to the OTHER groups' ranks idle.  They are used like this: the
// message s -> d is cut into a direct part and R = P - g relay stripes, stripe j travelling s -> r_j -> d through the
// j-th rank outside the group (two hops, staged in r_j's memory).  Every directed link between ranks of different groups
// then carries (g - 1) first-hop stripes and (g - 1) second-hop stripes, a link inside a group the direct part; with a
// stripe y = 1 / (2 (g - 1) + R) of the message and the direct part x = 2 (g - 1) y all links finish together:
//   8 ranks, 4 x 2 grid:  groups of 2 (R = 6): y = 1/8, x = 1/4  -> the busiest link carries 1/4 of what the single
//                         direct link carries today;  groups of 4 (R = 4): y = 1/10, x = 6/10.
// The exchange runs in two phases (one pull kernel each): phase 1 moves the first half of every direct part and all
// first hops, phase 2 the second half and all second hops.
//
// Layout of a message of `bytes`:  [direct, first half][direct, second half][stripe 0] ... [stripe R-1]
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <algorithm>
#include <vector>

namespace mfft {

struct RelayCut {
  int g = 1, R = 0;            // group size, relays per message
  size_t stripe = 0;           // bytes of every relay stripe (multiple of `gran`)
  size_t direct = 0;           // bytes of the direct part = bytes - R * stripe
  size_t half = 0;             // bytes of its first half (the second has direct - half)
  size_t stripe_off(int j) const { return direct + (size_t)j * stripe; }
};

// gran: stripes are whole multiples of it (cache lines / pages of the copy); a message too small to give every relay
// one granule goes direct as a whole
inline RelayCut relay_cut(size_t bytes, int g, int P, size_t gran = 4096) {
  RelayCut c;
  c.g = g;
  c.R = P - g;
  if (c.R <= 0 || g < 2) {
    c.R = 0;
    c.direct = bytes;
    c.half = bytes / 2 / 16 * 16;
    return c;
  }
  const size_t denom = (size_t)(2 * (g - 1) + c.R);
  c.stripe = bytes / denom / gran * gran;
  {
    // whole granules: of the two candidates around the exact stripe take the one whose busiest link carries less
    // (a link inside the group carries the direct part, one between groups 2 (g - 1) stripes)
    const size_t up = c.stripe + gran;
    auto worst = [&](size_t y) { return std::max(bytes - (size_t)c.R * y, (size_t)(2 * (g - 1)) * y); };
    if ((size_t)c.R * up <= bytes && worst(up) < worst(c.stripe)) c.stripe = up;
  }
  if (c.stripe == 0) c.R = 0;
  c.direct = bytes - (size_t)c.R * c.stripe;
  c.half = c.direct / 2 / 16 * 16;
  return c;
}

// the j-th relay of a message inside the group with id `gid`: the j-th rank (ascending) whose group id differs
inline int relay_rank(const int* part, int P, int gid, int j) {
  for (int r = 0; r < P; ++r)
    if (part[r] != gid && j-- == 0) return r;
  return -1;
}
// index of relay `r` among the ranks outside group `gid`, or -1
inline int relay_index(const int* part, int P, int gid, int r) {
  if (part[r] == gid) return -1;
  int j = 0;
  for (int q = 0; q < r; ++q)
    if (part[q] != gid) ++j;
  return j;
}

// Everything one rank PULLS in a relayed exchange, in the order the executor issues it.
//   kind 0: its own chunk (local copy, phase 1)
//   kind 1: direct part of the message msg_src -> rank, read from msg_src's send buffer (first half in phase 1, second in 2)
//   kind 2: first hop: this rank is a relay of msg_src -> msg_dst and reads its stripe from msg_src's send buffer into its
//           staging area (phase 1)
//   kind 3: second hop: stripe of msg_src -> rank read from the staging area of relay `from` (phase 2)
// msg_off is the offset inside the message; bytes_of(s, d) the size of the message s -> d (0: none), asked for every
// pair of every group.
struct RelayMove {
  int phase, kind, from, msg_src, msg_dst;
  size_t msg_off, bytes;
};
template <class BytesOf>
inline void relay_moves(int P, int rank, const int* part, BytesOf bytes_of, std::vector<RelayMove>* out) {
  const int gid = part[rank];
  int g = 0;
  for (int r = 0; r < P; ++r) g += part[r] == gid;
  out->clear();
  if (const size_t self = bytes_of(rank, rank)) out->push_back(RelayMove{1, 0, rank, rank, rank, 0, self});
  std::vector<RelayMove> p2;
  for (int s = 0; s < P; ++s) {               // as a destination
    if (s == rank || part[s] != gid) continue;
    const size_t b = bytes_of(s, rank);
    if (!b) continue;
    const RelayCut c = relay_cut(b, g, P);
    if (c.half) out->push_back(RelayMove{1, 1, s, s, rank, 0, c.half});
    if (c.direct > c.half) p2.push_back(RelayMove{2, 1, s, s, rank, c.half, c.direct - c.half});
    for (int j = 0; j < c.R; ++j) p2.push_back(RelayMove{2, 3, relay_rank(part, P, gid, j), s, rank, c.stripe_off(j), c.stripe});
  }
  for (int s = 0; s < P; ++s) {               // as a relay of the other groups' messages
    if (part[s] == gid) continue;
    const int j = relay_index(part, P, part[s], rank);
    for (int d = 0; d < P; ++d) {
      if (d == s || part[d] != part[s]) continue;
      const size_t b = bytes_of(s, d);
      if (!b) continue;
      int gs = 0;                              // the message is cut by ITS group's size (the destination cuts it the same way)
      for (int r = 0; r < P; ++r) gs += part[r] == part[s];
      const RelayCut c = relay_cut(b, gs, P);
      if (j < 0 || j >= c.R || !c.stripe) continue;
      out->push_back(RelayMove{1, 2, s, s, d, c.stripe_off(j), c.stripe});
    }
  }
  out->insert(out->end(), p2.begin(), p2.end());
}

}  // namespace mfft
