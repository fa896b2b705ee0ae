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
    } else if (kind == "cov") {
      // "cov <s1|s2> <fusion part> xx xy xz yy yz zz": one person with that joint at (1, 2, 3) and that covariance; prints every marker as
      // "ns id type lifetime alpha | px py pz | qx qy qz qw | sx sy sz" and, for s1, the running sigma range in mm
      std::string stage; int part; double c[6];
      in >> stage >> part;
      for (double& v : c) in >> v;
      person_msgs::PersonCovList l;
      person_msgs::PersonCov p;
      p.keypoints.resize(se3d::FU_NUM);
      p.keypoints[part].joint.x = 1; p.keypoints[part].joint.y = 2; p.keypoints[part].joint.z = 3; p.keypoints[part].score = 0.7f;
      for (int a = 0; a < 6; ++a) p.keypoints[part].cov[a] = c[a];
      l.persons.push_back(person_msgs::PersonCov());      // person 0 without joints: the ids must count from person 1
      l.persons.back().keypoints.resize(se3d::FU_NUM);
      l.persons.push_back(p);
      se3d::SigmaRange range;
      if (stage == "s1") se3d::covariance_markers_3d(l, "simple", arr, &range); else se3d::covariance_markers_fused(l, arr, 1.0);
      for (const auto& m : arr.markers)
        std::printf("%s %d %d %.1f %.2f | %.17g %.17g %.17g | %.17g %.17g %.17g %.17g | %.17g %.17g %.17g\n", m.ns.c_str(), m.id, m.type, m.lifetime, m.color.a,
                    m.position.x, m.position.y, m.position.z, m.orientation[0], m.orientation[1], m.orientation[2], m.orientation[3], m.scale[0], m.scale[1], m.scale[2]);
      if (stage == "s1" && !arr.markers.empty()) std::printf("range %.9g %.9g %.9g %.9g %.9g %.9g\n", range.min_mm(0), range.min_mm(1), range.min_mm(2), range.max_mm(0), range.max_mm(1), range.max_mm(2));
      std::printf("--\n");
      continue;
    } else continue;
    print(arr);
  }
  return 0;
}
