// se3d_markers_test.cpp — driver for tests/test_markers_cpu.py.  stdin: lines "s1 <pose_method> <mask hex>" (one stage-1 person whose fusion
// slots in the mask carry score 0.5), "s2 <track id> <stamp> <mask hex>" (one fused person; a line "s2 - <stamp> 0" publishes an empty list).
// Joint k sits at (k, 0, 0), so a printed point identifies its joint.  Output: one line per marker: ns id type action lifetime, then the points.
#include <cstdio>
#include <iostream>
#include <sstream>

#include "se3d_markers.hpp"

static person_msgs::PersonCov person_from_mask(unsigned mask, uint32_t id) {
  person_msgs::PersonCov p;
  p.id = id;
  p.keypoints.resize(se3d::FU_NUM);
  for (int k = 0; k < se3d::FU_NUM; ++k)
    if ((mask >> k) & 1u) { p.keypoints[k].joint.x = k; p.keypoints[k].score = 0.5f; }
  return p;
}
static void print(const visualization_msgs::MarkerArray& arr) {
  for (const auto& m : arr.markers) {
    std::printf("%s %d %d %d %.1f |", m.ns.c_str(), m.id, m.type, m.action, m.lifetime);
    for (const auto& pt : m.points) std::printf(" %d", (int)pt.x);
    std::printf(" | %zu colors, first %.4f %.4f %.4f\n", m.colors.size(), m.colors.empty() ? 0.f : m.colors[0].r, m.colors.empty() ? 0.f : m.colors[0].g,
                m.colors.empty() ? 0.f : m.colors[0].b);
  }
  std::printf("--\n");
}

int main() {
  se3d::FusedMarkerState state(1.0);
  std::string line;
  while (std::getline(std::cin, line)) {
    std::istringstream in(line);
    std::string kind;
    in >> kind;
    visualization_msgs::MarkerArray arr;
    if (kind == "s1") {
      std::string method; unsigned mask;
      in >> method >> std::hex >> mask;
      person_msgs::PersonCovList l;
      l.persons.push_back(person_from_mask(mask, 0));
      se3d::skeleton_markers(l, method, arr);
    } else if (kind == "s2") {
      std::string id; double stamp; unsigned mask;
      in >> id >> stamp >> std::hex >> mask;
      person_msgs::PersonCovList l;
      l.header.stamp = stamp;
      if (id != "-") l.persons.push_back(person_from_mask(mask, (uint32_t)std::stoul(id)));
      se3d::fused_markers(l, arr);
      state.update(l, arr);
    } else continue;
    print(arr);
  }
  return 0;
}
