// Drives FoldLine (include/zkhip_prover.hpp) through a simulated stream: leaf node k is finished at time leaf_gap * (k + 1) (+ jitter from
// a seeded generator when asked), a fold takes fold_time on one of `workers` workers, the number of leaf nodes is announced when the
// last one arrives.  Prints one line per fold: start end lo hi depth n_kids kinds... kid-ranges..., then "root lo hi depth end t_last".
// argv: m arity leaf_gap fold_time workers seed
#include <cstdio>
#include <cstdlib>
#include <queue>
#include <vector>

#include "zkhip_prover.hpp"

using scroll_zkvm_hip::FoldLine;

struct Tag {
    size_t lo, hi;
    int kind;
};

int main(int argc, char** argv) {
    if (argc < 7) return 2;
    const size_t m = strtoul(argv[1], nullptr, 10), arity = strtoul(argv[2], nullptr, 10), workers = strtoul(argv[5], nullptr, 10);
    const double leaf_gap = atof(argv[3]), fold_time = atof(argv[4]);
    unsigned seed = (unsigned)strtoul(argv[6], nullptr, 10);
    auto rnd = [&] {
        seed = seed * 1664525u + 1013904223u;
        return (seed >> 8) / (double)(1u << 24);
    };
    std::vector<double> arrive(m);
    for (size_t k = 0; k < m; k++) arrive[k] = leaf_gap * (k + 1) + (atoi(argv[6]) ? leaf_gap * 3 * rnd() : 0.0);   // (jitter: out of order)
    double t_last = 0;
    for (double a : arrive) t_last = a > t_last ? a : t_last;
    FoldLine line(arity);
    std::deque<Tag> tags;
    struct Running {
        double end;
        FoldLine::Fold f;
        bool operator<(const Running& o) const { return end > o.end; }
    };
    std::priority_queue<Running> running;
    std::deque<FoldLine::Fold> waiting;
    std::vector<std::pair<double, size_t>> order;
    for (size_t k = 0; k < m; k++) order.push_back({arrive[k], k});
    std::sort(order.begin(), order.end());
    size_t next_leaf = 0;
    double now = 0;
    auto pump = [&] {
        FoldLine::Fold f;
        while (line.next(&f)) waiting.push_back(f);
        while (!waiting.empty() && running.size() < workers) {
            FoldLine::Fold g = waiting.front();
            waiting.pop_front();
            std::printf("fold %.3f %.3f %zu %zu %zu %zu", now, now + fold_time, g.lo, g.hi, g.depth, g.kids.size());
            for (int kd : g.kinds) std::printf(" %d", kd);
            for (const void* t : g.kids) std::printf(" %zu %zu", ((const Tag*)t)->lo, ((const Tag*)t)->hi);
            std::printf("\n");
            running.push(Running{now + fold_time, g});
        }
    };
    while (!line.root()) {
        const double ta = next_leaf < m ? order[next_leaf].first : 1e300, tf = running.empty() ? 1e300 : running.top().end;
        if (ta == 1e300 && tf == 1e300) {
            pump();
            if (line.root()) break;
            if (running.empty() && waiting.empty()) {
                std::printf("stuck\n");
                return 1;
            }
            continue;
        }
        if (ta <= tf) {
            now = ta;
            const size_t k = order[next_leaf++].second;
            tags.push_back(Tag{k, k + 1, 1 + (int)(k % 2)});
            line.add(k, k + 1, tags.back().kind, 0, &tags.back());
            if (next_leaf == m) line.set_total(m);
        } else {
            now = tf;
            Running r = running.top();
            running.pop();
            tags.push_back(Tag{r.f.lo, r.f.hi, 0});
            line.done(r.f, &tags.back());
        }
        pump();
    }
    const Tag* r = (const Tag*)line.root();
    std::printf("root %zu %zu %zu %.3f %.3f %zu\n", r->lo, r->hi, line.root_depth(), now, t_last, line.folds());
    return 0;
}
