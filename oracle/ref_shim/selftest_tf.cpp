// TEST INFRASTRUCTURE ONLY — identities the restated tf / pcl_ros slice must satisfy (tests/test_roslog.py runs it). Not a pin
// against tf (unavailable here): a guard against slips in the restatement — wrong sign, transposed matrix, wrong direction.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <pcl_ros/transforms.h>
#include <tf/transform_broadcaster.h>
#include <tf/transform_listener.h>

static int fails = 0;
#define CHECK(c) do { if (!(c)) { std::printf("FAIL line %d: %s\n", __LINE__, #c); fails++; } } while (0)

int main() {
  std::srand(7);
  tf::TransformBroadcaster br; tf::TransformListener ls(ros::Duration(10));
  for (int it = 0; it < 2000; it++) {
    const double yaw = (std::rand() / (double)RAND_MAX - 0.5) * 12.0, tx = (std::rand() % 2000 - 1000) * 0.37, ty = (std::rand() % 2000 - 1000) * 0.11;
    tf::Quaternion q; q.setRPY(0, 0, yaw);
    CHECK(std::fabs(q.length2() - 1.0) < 1e-15 && q.x() == 0 && q.y() == 0);
    tf::Matrix3x3 m(q), e; e.setEulerYPR(yaw, 0, 0);
    const double c = std::cos(yaw), s = std::sin(yaw), want[3][3] = {{c, -s, 0}, {s, c, 0}, {0, 0, 1}};
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { CHECK(std::fabs(m[i][j] - want[i][j]) < 4e-16); CHECK(std::fabs(e[i][j] - want[i][j]) < 4e-16); }
    tf::Quaternion back; m.getRotation(back);
    const double sign = back.w() * q.w() + back.z() * q.z() < 0 ? -1.0 : 1.0;
    CHECK(std::fabs(sign * back.z() - q.z()) < 1e-15 && std::fabs(sign * back.w() - q.w()) < 1e-15);
    // the node's edge: parent "velodyne", child "global"; a point of the global frame expressed in velodyne = R p + t
    tf::Transform t; t.setOrigin(tf::Vector3(tx, ty, 0)); t.setRotation(q);
    br.sendTransform(tf::StampedTransform(t, ros::Time(1.0), "velodyne", "global"));
    pcl::PointCloud<pcl::PointXYZ> g, v, g2;
    g.header.frame_id = "global";
    g.push_back(pcl::PointXYZ(3.5f, -7.25f, 1.5f)); g.push_back(pcl::PointXYZ(-40.f, 12.f, -2.f));
    CHECK(ls.waitForTransform("/velodyne", "/global", ros::Time(), ros::Duration(1.0)));
    CHECK(pcl_ros::transformPointCloud("/velodyne", g, v, ls) && v.header.frame_id == "/velodyne");
    for (size_t i = 0; i < g.size(); i++) {
      const double x = c * g[i].x - s * g[i].y + tx, y = s * g[i].x + c * g[i].y + ty;
      const double tol = 4e-7 * (std::fabs(x) + std::fabs(y) + std::fabs(tx) + std::fabs(ty) + 60.0);
      CHECK(std::fabs(v[i].x - x) < tol && std::fabs(v[i].y - y) < tol && v[i].z == g[i].z);
    }
    v.header.frame_id = "velodyne";
    CHECK(pcl_ros::transformPointCloud("/global", v, g2, ls));     // and back: the inverse edge
    for (size_t i = 0; i < g.size(); i++) {
      const double tol = 1e-6 * (std::fabs(tx) + std::fabs(ty) + 60.0);
      CHECK(std::fabs(g2[i].x - g[i].x) < tol && std::fabs(g2[i].y - g[i].y) < tol && g2[i].z == g[i].z);
    }
  }
  ros::Time a(1000.25); CHECK(a.sec == 1000 && a.nsec == 250000000 && a.toSec() == 1000.25);
  ros::Duration d(0.1); CHECK(d.sec == 0 && d.nsec == 100000000);
  ros::Time n; n.fromNSec(1234567891234ull); CHECK(n.sec == 1234 && n.nsec == 567891234 && n.toNSec() == 1234567891234ull);
  std::printf(fails ? "%d failures\n" : "ok\n", fails);
  return fails != 0;
}
