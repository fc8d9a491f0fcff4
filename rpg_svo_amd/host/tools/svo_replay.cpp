// svo_replay -- ROS-free dataset replay through svo::FrameHandlerMono.
//
// The role of svo_ros/src/benchmark_node.cpp:91-131,178-256 without ROS: read a dataset in the
// layout of the reference's Blender sequences (trajectory.txt, img/<name>_0.png,
// depth/<name>_0.depth), start the pipeline from the first frame at its ground-truth pose with FAST
// corners lifted through the depth map (benchmark_node.cpp:216-235), feed every further image through
// FrameHandlerMono::addImage, and write what the reference's benchmark writes:
//
//   <out>/traj_estimate.txt   timestamp tx ty tz qx qy qz qw      (T_world_from_frame, :118-131)
//   <out>/svo.csv             the per-frame trace FrameHandlerBase keeps itself when built with
//                             -DSVO_TRACE (frame_handler_base.cpp:46-74): same columns as the reference
//
// It is an ordinary client of the reference's classes: linked against the reference's own libsvo it
// is the CPU benchmark, linked against the drop-in bodies (rpg_svo_amd/host/dropin/*.cpp +
// libsvo_hip.so) it runs the hot path on the MI355X.  No arithmetic of the path lives here.
//
//   svo_replay --dataset DIR --out DIR --cam pinhole:W,H,fx,fy,cx,cy[,d0,d1,d2,d3]
//              [--cam atan:W,H,fx,fy,cx,cy,s]  [--frames N] [--pyr-levels 3] [--max-fts 120]
#include <zlib.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include <svo/config.h>
#include <svo/depth_filter.h>
#include <svo/feature.h>
#include <svo/feature_detection.h>
#include <svo/frame.h>
#include <svo/frame_handler_mono.h>
#include <svo/map.h>
#include <svo/point.h>
#include <vikit/atan_camera.h>
#include <vikit/pinhole_camera.h>

using namespace svo;

namespace {

struct Entry {
  double timestamp;
  std::string name;
  double t[3], q[4];  // T_world_from_frame: translation, quaternion (x, y, z, w)
};

std::vector<Entry> readTrajectory(const std::string& dir) {
  std::ifstream in((dir + "/trajectory.txt").c_str());
  if (!in) throw std::runtime_error("cannot open " + dir + "/trajectory.txt");
  std::vector<Entry> out;
  std::string line;
  while (std::getline(in, line)) {
    if (line.empty() || line[0] == '#') continue;
    std::istringstream ss(line);
    Entry e;
    if (ss >> e.timestamp >> e.name >> e.t[0] >> e.t[1] >> e.t[2] >> e.q[0] >> e.q[1] >> e.q[2] >> e.q[3]) out.push_back(e);
  }
  return out;
}

// 8-bit grayscale, non-interlaced PNG (what the datasets hold); all five row filters
std::vector<uint8_t> readPngGray8(const std::string& path, int& w, int& h) {
  std::ifstream in(path.c_str(), std::ios::binary);
  if (!in) throw std::runtime_error("cannot open " + path);
  std::vector<uint8_t> d((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
  static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
  if (d.size() < 8 || std::memcmp(&d[0], sig, 8) != 0) throw std::runtime_error(path + ": not a PNG file");
  size_t pos = 8;
  std::vector<uint8_t> idat;
  int depth = 0, ctype = -1, interlace = 0;
  w = h = 0;
  while (pos + 8 <= d.size()) {
    const uint32_t n = (uint32_t(d[pos]) << 24) | (uint32_t(d[pos + 1]) << 16) | (uint32_t(d[pos + 2]) << 8) | d[pos + 3];
    const std::string tag(reinterpret_cast<const char*>(&d[pos + 4]), 4);
    const uint8_t* body = &d[pos + 8];
    if (tag == "IHDR") {
      w = (body[0] << 24) | (body[1] << 16) | (body[2] << 8) | body[3];
      h = (body[4] << 24) | (body[5] << 16) | (body[6] << 8) | body[7];
      depth = body[8]; ctype = body[9]; interlace = body[12];
    } else if (tag == "IDAT") {
      idat.insert(idat.end(), body, body + n);
    } else if (tag == "IEND") {
      break;
    }
    pos += 12 + n;
  }
  if (depth != 8 || ctype != 0 || interlace != 0) throw std::runtime_error(path + ": only 8-bit grayscale non-interlaced PNG");
  std::vector<uint8_t> raw((size_t)h * (w + 1));
  uLongf len = raw.size();
  if (uncompress(&raw[0], &len, &idat[0], idat.size()) != Z_OK || len != raw.size()) throw std::runtime_error(path + ": inflate failed");
  std::vector<uint8_t> img((size_t)w * h);
  for (int y = 0; y < h; ++y) {
    const uint8_t f = raw[(size_t)y * (w + 1)];
    const uint8_t* src = &raw[(size_t)y * (w + 1) + 1];
    uint8_t* dst = &img[(size_t)y * w];
    const uint8_t* up = y ? dst - w : NULL;
    for (int x = 0; x < w; ++x) {
      const int a = x ? dst[x - 1] : 0, b = up ? up[x] : 0, c = (x && up) ? up[x - 1] : 0;
      int pred = 0;
      if (f == 1) pred = a;
      else if (f == 2) pred = b;
      else if (f == 3) pred = (a + b) / 2;
      else if (f == 4) {
        const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
        pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
      }
      dst[x] = (uint8_t)(src[x] + pred);
    }
  }
  return img;
}

// vk::blender_utils::loadBlenderDepthmap: z-depths, row-major; turned into range along the viewing ray
std::vector<float> readRangeMap(const std::string& path, const vk::AbstractCamera& cam) {
  std::ifstream in(path.c_str());
  if (!in) throw std::runtime_error("cannot open " + path);
  const int w = cam.width(), h = cam.height();
  std::vector<float> r((size_t)w * h);
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) {
      double z;
      if (!(in >> z)) throw std::runtime_error(path + ": too few depth values");
      const Vector3d f = cam.cam2world(x, y);
      r[(size_t)y * w + x] = (float)(z / f[2]);
    }
  return r;
}

SE3 poseWorldFromFrame(const Entry& e) {
  const double x = e.q[0], y = e.q[1], z = e.q[2], w = e.q[3];
  Matrix3d R;
  R(0, 0) = 1 - 2 * (y * y + z * z); R(0, 1) = 2 * (x * y - z * w);     R(0, 2) = 2 * (x * z + y * w);
  R(1, 0) = 2 * (x * y + z * w);     R(1, 1) = 1 - 2 * (x * x + z * z); R(1, 2) = 2 * (y * z - x * w);
  R(2, 0) = 2 * (x * z - y * w);     R(2, 1) = 2 * (y * z + x * w);     R(2, 2) = 1 - 2 * (x * x + y * y);
  return SE3(R, Vector3d(e.t[0], e.t[1], e.t[2]));
}

void quatOf(const Matrix3d& R, double q[4]) {  // (x, y, z, w), w >= 0
  const double tr = R(0, 0) + R(1, 1) + R(2, 2);
  double w, x, y, z;
  if (tr > 0) {
    const double s = std::sqrt(tr + 1.0) * 2;
    w = 0.25 * s; x = (R(2, 1) - R(1, 2)) / s; y = (R(0, 2) - R(2, 0)) / s; z = (R(1, 0) - R(0, 1)) / s;
  } else if (R(0, 0) > R(1, 1) && R(0, 0) > R(2, 2)) {
    const double s = std::sqrt(1.0 + R(0, 0) - R(1, 1) - R(2, 2)) * 2;
    w = (R(2, 1) - R(1, 2)) / s; x = 0.25 * s; y = (R(0, 1) + R(1, 0)) / s; z = (R(0, 2) + R(2, 0)) / s;
  } else if (R(1, 1) > R(2, 2)) {
    const double s = std::sqrt(1.0 + R(1, 1) - R(0, 0) - R(2, 2)) * 2;
    w = (R(0, 2) - R(2, 0)) / s; x = (R(0, 1) + R(1, 0)) / s; y = 0.25 * s; z = (R(1, 2) + R(2, 1)) / s;
  } else {
    const double s = std::sqrt(1.0 + R(2, 2) - R(0, 0) - R(1, 1)) * 2;
    w = (R(1, 0) - R(0, 1)) / s; x = (R(0, 2) + R(2, 0)) / s; y = (R(1, 2) + R(2, 1)) / s; z = 0.25 * s;
  }
  if (w < 0) { w = -w; x = -x; y = -y; z = -z; }
  q[0] = x; q[1] = y; q[2] = z; q[3] = w;
}

vk::AbstractCamera* makeCamera(const std::string& spec) {
  const size_t colon = spec.find(':');
  if (colon == std::string::npos) throw std::runtime_error("--cam wants model:W,H,fx,fy,cx,cy[,...]");
  const std::string model = spec.substr(0, colon);
  std::vector<double> v;
  std::stringstream ss(spec.substr(colon + 1));
  std::string tok;
  while (std::getline(ss, tok, ',')) v.push_back(std::atof(tok.c_str()));
  if (model == "atan" && v.size() == 7) return new vk::ATANCamera(v[0], v[1], v[2], v[3], v[4], v[5], v[6]);
  if (model == "pinhole" && v.size() >= 6) {
    v.resize(11, 0.0);
    return new vk::PinholeCamera(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], v[8], v[9], v[10]);
  }
  throw std::runtime_error("--cam: pinhole:W,H,fx,fy,cx,cy[,d0,d1,d2,d3[,d4]] or atan:W,H,fx,fy,cx,cy,s");
}

}  // namespace

int main(int argc, char** argv) {
  std::string dataset, out, cam_spec = "pinhole:752,480,315.5,315.5,376,240";
  int n_frames = -1;
  bool mapper_thread = true;  // the reference's mode: DepthFilter runs in its own thread (depth_filter.cpp:70-76)
  for (int i = 1; i + 1 < argc; i += 2) {
    const std::string k = argv[i], v = argv[i + 1];
    if (k == "--dataset") dataset = v;
    else if (k == "--out") out = v;
    else if (k == "--cam") cam_spec = v;
    else if (k == "--frames") n_frames = std::atoi(v.c_str());
    else if (k == "--pyr-levels") Config::nPyrLevels() = std::atoi(v.c_str());
    else if (k == "--max-fts") Config::maxFts() = std::atoi(v.c_str());
    else if (k == "--kfselect-mindist") Config::kfSelectMinDist() = std::atof(v.c_str());
    else if (k == "--mapper-thread") mapper_thread = std::atoi(v.c_str()) != 0;
    else { std::fprintf(stderr, "unknown option %s\n", k.c_str()); return 2; }
  }
  if (dataset.empty() || out.empty()) {
    std::fprintf(stderr, "usage: svo_replay --dataset DIR --out DIR [--cam model:params] [--frames N] [--mapper-thread 0|1]\n");
    return 2;
  }
  try {
    vk::AbstractCamera* cam = makeCamera(cam_spec);
    std::vector<Entry> traj = readTrajectory(dataset);
    if (n_frames > 0 && (size_t)n_frames < traj.size()) traj.resize(n_frames);
    if (traj.size() < 2) throw std::runtime_error("dataset has fewer than two frames");
    Config::traceDir() = out;
    Config::traceName() = "svo";
    std::srand(1);  // Reprojector::initializeGrid's random_shuffle (reprojector.cpp:54)
    FrameHandlerMono vo(cam);
    vo.start();
    // --mapper-thread 0: the depth filter updates its seeds inside addImage (DepthFilter::addFrame without a thread,
    // depth_filter.cpp:97-110): a replay is then deterministic -- which frame first sees a converged seed's point no longer
    // depends on how far the mapper thread got
    if (!mapper_thread) vo.depthFilter()->stopThread();
    std::ofstream est((out + "/traj_estimate.txt").c_str());
    est.precision(12);
    size_t n_tracked = 0;
    for (size_t i = 0; i < traj.size(); ++i) {
      int w = 0, h = 0;
      std::vector<uint8_t> px = readPngGray8(dataset + "/img/" + traj[i].name + "_0.png", w, h);
      if (w != cam->width() || h != cam->height()) throw std::runtime_error("image size does not match the camera");
      cv::Mat img(h, w, CV_8UC1);
      std::memcpy(img.data, &px[0], px.size());
      if (i == 0) {
        // benchmark_node.cpp:216-235: first frame at its ground-truth pose, corners lifted through the depth map
        FramePtr ref(new Frame(cam, img, traj[0].timestamp));
        ref->T_f_w_ = poseWorldFromFrame(traj[0]).inverse();
        const std::vector<float> range = readRangeMap(dataset + "/depth/" + traj[0].name + "_0.depth", *cam);
        feature_detection::FastDetector detector(w, h, Config::gridSize(), Config::nPyrLevels());
        detector.detect(ref.get(), ref->img_pyr_, Config::triangMinCornerScore(), ref->fts_);
        for (Features::iterator it = ref->fts_.begin(); it != ref->fts_.end(); ++it) {
          Feature* ftr = *it;
          const Vector3d p_cam = ftr->f * (double)range[(size_t)((int)ftr->px[1]) * w + (int)ftr->px[0]];
          Point* point = new Point(ref->T_f_w_.inverse() * p_cam, ftr);
          ftr->point = point;
        }
        vo.setFirstFrame(ref);
      } else {
        vo.addImage(img, traj[i].timestamp);
      }
      FramePtr f = vo.lastFrame();
      if (f && (i == 0 || vo.stage() == FrameHandlerBase::STAGE_DEFAULT_FRAME)) {
        const SE3 T_w_f = f->T_f_w_.inverse();
        double q[4];
        quatOf(T_w_f.rotation_matrix(), q);
        const Vector3d t = T_w_f.translation();
        est << traj[i].timestamp << " " << t[0] << " " << t[1] << " " << t[2] << " " << q[0] << " " << q[1] << " " << q[2]
            << " " << q[3] << "\n";
        ++n_tracked;
      }
    }
    est.close();
    std::printf("svo_replay: %zu of %zu frames tracked, %zu keyframes; wrote %s/traj_estimate.txt and %s/svo.csv\n", n_tracked,
                traj.size(), vo.map().size(), out.c_str(), out.c_str());
    return n_tracked == traj.size() ? 0 : 1;
  } catch (const std::exception& e) {
    std::fprintf(stderr, "svo_replay: %s\n", e.what());
    return 3;
  }
}
