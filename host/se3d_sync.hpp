// se3d_sync.hpp — ROS-free restatement of the reference's runtime-sized approximate-time synchroniser
// (skeleton_3d/include/my_message_filters/sync_policies/approximate_time_vec.h, "SYNC"; wiring in
// my_message_filters/synchronizer_vec.h, "SYNCV", and S3D:1218-1222).  It sits immediately before the accelerated path: it turns
// the N per-camera Person2DList streams into the N-tuples that skeletonCallback / skeletonThreadCallback consume.
//
// Behaviour restated (not the code): one FIFO per input plus a "past" stack of messages that were already considered for the
// current pivot; a candidate tuple is the set of FIFO heads; the pivot is the input whose head is the latest when the first
// candidate is formed; candidates are improved while (end - candidate_end) * (1 + age_penalty) < (start - candidate_start); the
// best candidate is published once the pivot is exhausted or optimality is proven — if some FIFO is empty, optimistically, using
// the per-input lower bound on the message period ("virtual" heads); a FIFO that overflows drops its oldest message, which
// invalidates the candidate and bars that input from becoming pivot until it is no longer the latest head.
// The node uses queue_size = max(3, 1 + N/4), inter-message lower bound 20 ms, age penalty 2.0 (S3D:1219-1221).
//
// Times are integer nanoseconds.  ros::Duration * double goes through seconds as a double and rounds back to nanoseconds
// (rostime DurationBase::fromSec); scaled() reproduces that so that ties resolve the same way.
#pragma once
#include <cmath>
#include <cstdint>
#include <deque>
#include <functional>
#include <limits>
#include <memory>
#include <vector>

namespace se3d {

template <typename Msg>
class ApproximateTimeSync {
 public:
  using MsgPtr = std::shared_ptr<const Msg>;
  using Callback = std::function<void(const std::vector<MsgPtr>&)>;
  using StampFn = std::function<int64_t(const Msg&)>;  // header.stamp in nanoseconds

  ApproximateTimeSync(uint32_t queue_size, int num_inputs, StampFn stamp)
      : queue_size_(queue_size), n_(num_inputs), stamp_(std::move(stamp)), in_(num_inputs), candidate_(num_inputs) {}

  void registerCallback(Callback cb) { cb_ = std::move(cb); }
  void setAgePenalty(double p) { age_penalty_ = p; }                                    // SYNC:217-222
  void setInterMessageLowerBound(int64_t ns) { for (auto& q : in_) q.lower_bound = ns; }  // SYNC:229-235
  void setInterMessageLowerBound(int i, int64_t ns) { in_[i].lower_bound = ns; }
  void setMaxIntervalDuration(int64_t ns) { max_interval_ = ns; }
  int warnings() const { return warnings_; }  // out-of-order / too-close messages seen (the reference prints one warning per input)

  // SYNC:170-215
  void add(MsgPtr m, int i) {
    Input& q = in_[i];
    q.fifo.push_back(std::move(m));
    if (q.fifo.size() == 1) {
      if (++non_empty_ == (uint32_t)n_) process();
    } else {
      checkBound(i);
    }
    if (q.fifo.size() + q.past.size() > queue_size_) {  // overflow: cancel the search, drop the oldest message of this input
      non_empty_ = 0;
      for (int j = 0; j < n_; ++j) restoreAll(j);
      q.fifo.pop_front();
      q.dropped = true;
      if (pivot_ >= 0) {
        clearCandidate();
        process();
      }
    }
  }

 private:
  struct Input {
    std::deque<MsgPtr> fifo;
    std::vector<MsgPtr> past;
    bool dropped = false, warned = false;
    int64_t lower_bound = 0;
  };

  int64_t t(const MsgPtr& m) const { return stamp_(*m); }
  static int64_t scaled(int64_t ns, double scale) {  // ros::Duration * double
    const int64_t s = ns / 1000000000ll - ((ns % 1000000000ll) < 0 ? 1 : 0);
    const int64_t r = ns - s * 1000000000ll;
    const double d = ((double)s + 1e-9 * (double)r) * scale;
    const double fl = std::floor(d);
    return (int64_t)fl * 1000000000ll + (int64_t)std::llround((d - fl) * 1e9);
  }
  void clearCandidate() {
    for (auto& c : candidate_) c.reset();
    pivot_ = -1;
  }
  void checkBound(int i) {  // SYNC:129-168
    Input& q = in_[i];
    if (q.warned) return;
    const int64_t now = t(q.fifo.back());
    int64_t prev;
    if (q.fifo.size() == 1) {
      if (q.past.empty()) return;
      prev = t(q.past.back());
    } else {
      prev = t(q.fifo[q.fifo.size() - 2]);
    }
    if (now < prev || now - prev < q.lower_bound) { q.warned = true; ++warnings_; }
  }
  void popFront(int i) {  // SYNC:243-257
    in_[i].fifo.pop_front();
    if (in_[i].fifo.empty()) --non_empty_;
  }
  void frontToPast(int i) {  // SYNC:260-276
    in_[i].past.push_back(in_[i].fifo.front());
    popFront(i);
  }
  void restore(size_t count, int i) {  // SYNC:296-318
    Input& q = in_[i];
    while (count-- > 0) { q.fifo.push_front(q.past.back()); q.past.pop_back(); }
    if (!q.fifo.empty()) ++non_empty_;
  }
  void restoreAll(int i) { restore(in_[i].past.size(), i); }  // SYNC:321-341
  void takeCandidate() {  // SYNC:278-291
    for (int i = 0; i < n_; ++i) candidate_[i] = in_[i].fifo.front();
    for (auto& q : in_) q.past.clear();
  }
  void publish() {  // SYNC:369-384, SYNCV:147-161
    if (cb_) cb_(candidate_);
    clearCandidate();
    non_empty_ = 0;
    for (int i = 0; i < n_; ++i) {  // bring back the hidden messages and remove the published one
      Input& q = in_[i];
      while (!q.past.empty()) { q.fifo.push_front(q.past.back()); q.past.pop_back(); }
      q.fifo.pop_front();
      if (!q.fifo.empty()) ++non_empty_;
    }
  }
  // earliest (end = false: ties keep the lowest input index) or latest (end = true: ties go to the highest index) head (SYNC:405-421)
  void boundary(bool end, int& index, int64_t& time) const {
    time = t(in_[0].fifo.front());
    index = 0;
    for (int i = 1; i < n_; ++i) {
      const int64_t ti = t(in_[i].fifo.front());
      if ((ti < time) != end) { time = ti; index = i; }
    }
  }
  int64_t virtualTime(int i) const {  // SYNC:425-451
    const Input& q = in_[i];
    if (q.fifo.empty()) {
      const int64_t lb = t(q.past.back()) + q.lower_bound;
      return lb > pivot_time_ ? lb : pivot_time_;
    }
    return t(q.fifo.front());
  }
  void virtualBoundary(bool end, int& index, int64_t& time) const {  // SYNC:469-489
    time = virtualTime(0);
    index = 0;
    for (int i = 0; i < n_; ++i) {
      const int64_t ti = virtualTime(i);
      if ((ti < time) != end) { time = ti; index = i; }
    }
  }
  bool notBetter(int64_t end, int64_t start_ref) const {  // (end - candidate_end) * (1 + penalty) >= start_ref - candidate_start
    return scaled(end - cand_end_, 1.0 + age_penalty_) >= start_ref - cand_start_;
  }

  void process() {  // SYNC:493-626
    while (non_empty_ == (uint32_t)n_) {
      int e_idx, s_idx;
      int64_t e_t, s_t;
      boundary(true, e_idx, e_t);
      boundary(false, s_idx, s_t);
      for (int i = 0; i < n_; ++i)
        if (i != e_idx) in_[i].dropped = false;
      if (pivot_ < 0) {
        if (e_t - s_t > max_interval_ || in_[e_idx].dropped) {  // not a valid candidate / not a valid pivot
          popFront(s_idx);
          continue;
        }
        takeCandidate();
        cand_start_ = s_t;
        cand_end_ = e_t;
        pivot_ = e_idx;
        pivot_time_ = e_t;
        frontToPast(s_idx);
      } else if (notBetter(e_t, s_t)) {
        frontToPast(s_idx);
      } else {
        takeCandidate();
        cand_start_ = s_t;
        cand_end_ = e_t;
        frontToPast(s_idx);
      }
      if (s_idx == pivot_) {
        publish();  // all candidates of this pivot have been seen
      } else if (notBetter(e_t, pivot_time_)) {
        publish();  // provably optimal: every later candidate contains [pivot_time, end]
      } else if (non_empty_ < (uint32_t)n_) {
        // some FIFO ran empty: try to prove optimality with the rate bounds (virtual heads)
        std::vector<size_t> moved(n_, 0);
        for (;;) {
          int ve_idx, vs_idx;
          int64_t ve_t, vs_t;
          virtualBoundary(true, ve_idx, ve_t);
          virtualBoundary(false, vs_idx, vs_t);
          if (notBetter(ve_t, pivot_time_)) {
            publish();
            break;
          }
          if (!notBetter(ve_t, vs_t)) {  // an optimistic candidate would be better: wait for more messages
            non_empty_ = 0;
            for (int i = 0; i < n_; ++i) restore(moved[i], i);
            break;
          }
          frontToPast(vs_idx);
          ++moved[vs_idx];
        }
      }
    }
  }

  uint32_t queue_size_;
  int n_;
  StampFn stamp_;
  std::vector<Input> in_;
  std::vector<MsgPtr> candidate_;
  Callback cb_;
  uint32_t non_empty_ = 0;
  int pivot_ = -1;
  int64_t pivot_time_ = 0, cand_start_ = 0, cand_end_ = 0;
  int64_t max_interval_ = std::numeric_limits<int64_t>::max();
  double age_penalty_ = 0.1;
  int warnings_ = 0;
};

}  // namespace se3d
