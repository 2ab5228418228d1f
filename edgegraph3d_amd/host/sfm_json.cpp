// OpenMVG-JSON in/out for the hot path's host side (SURVEY row a-IO).
//
// Behaviour reproduced: OpenMvgParser::parse* (reference external/manifoldReconstructor/src/
// OpenMvgParser.cpp:39-301: view index = position in `views`, observation keys mapped through the
// position of the pose in `extrinsics`, t = -R*C, P = K4*[R t;0 1], distortion ignored) and
// output_sfm_data (src/edgegraph3d/io/output/output_sfm_data.cpp:186-229: sfm_data_version,
// root_path, views, intrinsics, control_points copied from the input file; extrinsics rewritten
// with keys 0..V-1; structure rewritten with keys 0..N-1 and id_feat 0). Own design: a small
// recursive-descent JSON DOM instead of rapidjson. The TEXT of the written file is what rapidjson's
// PrettyWriter produces for the same document (json_text.hpp): floats through Grisu2 + its notation rule,
// numbers copied from the input re-printed from the reader's verdict on them, strings decoded and re-escaped —
// checked byte for byte against the reference tree's vendored rapidjson (tests/test_json_rapidjson.py).
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

#ifdef _OPENMP
#include <omp.h>
#endif

#include "../../include/eg3d_host.h"
#include "camera_model.hpp"
#include "json_text.hpp"

namespace {

struct JVal {
  enum T { NUL, BOOL, NUM, STR, ARR, OBJ } t = NUL;
  bool b = false;
  std::string s;  // string value, or the source text of a number
  std::vector<JVal> a;
  std::vector<std::pair<std::string, JVal>> o;
  const JVal* get(const char* k) const {
    for (auto& kv : o)
      if (kv.first == k) return &kv.second;
    return nullptr;
  }
  double num() const { return t == NUM ? eg3d_json::parse_double(s) : 0.0; }
};

struct Parser {
  const char* p;
  const char* e;
  bool ok = true;
  int depth = 0;  // nesting of the value being parsed (a hostile file must not overflow the stack)
  void ws() {
    while (p < e && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) p++;
  }
  JVal parse() {
    ws();
    JVal v;
    if (p >= e || depth > 64) {
      ok = false;
      return v;
    }
    struct Depth {
      int& d;
      explicit Depth(int& x) : d(x) { d++; }
      ~Depth() { d--; }
    } guard(depth);
    if (*p == '{') {
      v.t = JVal::OBJ;
      p++;
      ws();
      if (p < e && *p == '}') {
        p++;
        return v;
      }
      while (ok) {
        ws();
        JVal k = parse_string();
        ws();
        if (p >= e || *p != ':') {
          ok = false;
          break;
        }
        p++;
        v.o.emplace_back(k.s, parse());
        ws();
        if (p < e && *p == ',') {
          p++;
          continue;
        }
        if (p < e && *p == '}') {
          p++;
          break;
        }
        ok = false;
      }
    } else if (*p == '[') {
      v.t = JVal::ARR;
      p++;
      ws();
      if (p < e && *p == ']') {
        p++;
        return v;
      }
      while (ok) {
        v.a.push_back(parse());
        ws();
        if (p < e && *p == ',') {
          p++;
          continue;
        }
        if (p < e && *p == ']') {
          p++;
          break;
        }
        ok = false;
      }
    } else if (*p == '"') {
      v = parse_string();
    } else if (!strncmp(p, "true", 4)) {
      v.t = JVal::BOOL;
      v.b = true;
      p += 4;
    } else if (!strncmp(p, "false", 5)) {
      v.t = JVal::BOOL;
      p += 5;
    } else if (!strncmp(p, "null", 4)) {
      p += 4;
    } else {
      const char* q = p;
      while (p < e && (strchr("+-0123456789.eE", *p))) p++;
      if (p == q) ok = false;
      v.t = JVal::NUM;
      v.s.assign(q, p);
    }
    return v;
  }
  JVal parse_string() {
    JVal v;
    v.t = JVal::STR;
    if (p >= e || *p != '"') {
      ok = false;
      return v;
    }
    p++;
    auto hex4 = [&](uint32_t& out) {
      out = 0;
      for (int k = 0; k < 4; k++) {
        if (p >= e) return false;
        const char c = *p++;
        out <<= 4;
        if (c >= '0' && c <= '9')
          out |= (uint32_t)(c - '0');
        else if (c >= 'a' && c <= 'f')
          out |= (uint32_t)(c - 'a' + 10);
        else if (c >= 'A' && c <= 'F')
          out |= (uint32_t)(c - 'A' + 10);
        else
          return false;
      }
      return true;
    };
    while (p < e && *p != '"') {  // the value is kept DECODED (the writer re-escapes it its own way)
      if (*p == '\\') {
        p++;
        if (p >= e) break;
        const char c = *p++;
        switch (c) {
          case '"': v.s += '"'; break;
          case '\\': v.s += '\\'; break;
          case '/': v.s += '/'; break;
          case 'b': v.s += '\b'; break;
          case 'f': v.s += '\f'; break;
          case 'n': v.s += '\n'; break;
          case 'r': v.s += '\r'; break;
          case 't': v.s += '\t'; break;
          case 'u': {
            uint32_t cp = 0, lo = 0;
            if (!hex4(cp)) {
              ok = false;
              return v;
            }
            if (cp >= 0xD800 && cp <= 0xDBFF) {  // surrogate pair
              if (p + 1 < e && p[0] == '\\' && p[1] == 'u') {
                p += 2;
                if (!hex4(lo) || lo < 0xDC00 || lo > 0xDFFF) {
                  ok = false;
                  return v;
                }
                cp = 0x10000 + (((cp - 0xD800) << 10) | (lo - 0xDC00));
              } else {
                ok = false;
                return v;
              }
            }
            eg3d_json::append_utf8(v.s, cp);
            break;
          }
          default: ok = false; return v;
        }
      } else if ((unsigned char)*p < 0x20) {
        // a raw control character inside a string is not JSON (RFC 4627; the reference's reader rejects it too,
        // external/rapidjson/reader.h:869)
        ok = false;
        return v;
      } else
        v.s.push_back(*p++);
    }
    if (p < e)
      p++;
    else
      ok = false;
    return v;
  }
};

void write_val(std::ostream& os, const JVal& v, int ind);
void indent(std::ostream& os, int n) {
  for (int i = 0; i < n; i++) os << "    ";
}
void write_val(std::ostream& os, const JVal& v, int ind) {
  switch (v.t) {
    case JVal::NUL: os << "null"; break;
    case JVal::BOOL: os << (v.b ? "true" : "false"); break;
    case JVal::NUM: {
      // a number copied through from the input: what the reference's reader + writer make of the literal
      bool good = true;
      const std::string n = eg3d_json::normalize_number(v.s, &good);
      os << (good ? n : v.s);
      break;
    }
    case JVal::STR: os << '"' << eg3d_json::escape_string(v.s) << '"'; break;
    case JVal::ARR:
      if (v.a.empty()) {
        os << "[]";
        break;
      }
      os << "[\n";
      for (size_t i = 0; i < v.a.size(); i++) {
        indent(os, ind + 1);
        write_val(os, v.a[i], ind + 1);
        os << (i + 1 < v.a.size() ? ",\n" : "\n");
      }
      indent(os, ind);
      os << "]";
      break;
    case JVal::OBJ:
      if (v.o.empty()) {
        os << "{}";
        break;
      }
      os << "{\n";
      for (size_t i = 0; i < v.o.size(); i++) {
        indent(os, ind + 1);
        os << '"' << eg3d_json::escape_string(v.o[i].first) << "\": ";
        write_val(os, v.o[i].second, ind + 1);
        os << (i + 1 < v.o.size() ? ",\n" : "\n");
      }
      indent(os, ind);
      os << "}";
      break;
  }
}

// shortest decimal text that round-trips a float widened to double (rapidjson writes doubles)
// a NaN / infinity has no JSON text (rapidjson's writer refuses it): the writer emits null and
// eg3d_sfm_write_json reports the file as invalid (-2)
thread_local bool g_nonfinite_written = false;
std::string num_text(float f) {
  double d = (double)f;
  if (!(d == d) || d > 1.7e308 || d < -1.7e308) {
    g_nonfinite_written = true;
    return "null";
  }
  return eg3d_json::double_text(d);
}
JVal jnum(float f) {
  JVal v;
  v.t = JVal::NUM;
  v.s = num_text(f);
  return v;
}
JVal jint(long long i) {
  JVal v;
  v.t = JVal::NUM;
  v.s = std::to_string(i);
  return v;
}

struct Cam {
  float focal = 0, ppx = 0, ppy = 0;
  float R[9] = {0}, C[3] = {0}, t[3] = {0}, P[16] = {0};
  std::string path;
};

}  // namespace

struct eg3d_sfm {
  int width = 0, height = 0;
  std::vector<Cam> cams;
  std::vector<float> P;  // flattened
  std::vector<float> X;  // [N][3]
  std::vector<uint32_t> trk_off{0};
  std::vector<int32_t> trk_view;
  std::vector<float> trk_xy;
  void refresh_P() {
    P.resize(cams.size() * 16);
    for (size_t i = 0; i < cams.size(); i++) memcpy(&P[i * 16], cams[i].P, sizeof(float) * 16);
  }
};

extern "C" eg3d_sfm* eg3d_sfm_create(int n_views, int width, int height) {
  eg3d_sfm* s = new eg3d_sfm();
  s->cams.resize(n_views);
  s->width = width;
  s->height = height;
  s->refresh_P();
  return s;
}
extern "C" void eg3d_sfm_destroy(eg3d_sfm* s) { delete s; }
extern "C" int eg3d_sfm_n_views(const eg3d_sfm* s) { return (int)s->cams.size(); }
extern "C" uint64_t eg3d_sfm_n_points(const eg3d_sfm* s) { return s->X.size() / 3; }
extern "C" const float* eg3d_sfm_cam_P(const eg3d_sfm* s) { return s->P.data(); }
extern "C" const float* eg3d_sfm_points(const eg3d_sfm* s) { return s->X.data(); }

extern "C" int eg3d_sfm_set_camera(eg3d_sfm* s, int view, float focal, float ppx, float ppy, const float* R9,
                                   const float* C3, const char* image_path) {
  if (!s || view < 0 || view >= (int)s->cams.size()) return -1;
  Cam& c = s->cams[view];
  c.focal = focal;
  c.ppx = ppx;
  c.ppy = ppy;
  memcpy(c.R, R9, sizeof(float) * 9);
  memcpy(c.C, C3, sizeof(float) * 3);
  if (image_path) c.path = image_path;
  eg3dh::translation_from_center(c.R, c.C, c.t);
  eg3dh::camera_matrix(focal, ppx, ppy, c.R, c.t, c.P);
  s->refresh_P();
  return 0;
}

extern "C" const char* eg3d_sfm_image_path(const eg3d_sfm* s, int view) {
  if (!s || view < 0 || view >= (int)s->cams.size()) return nullptr;
  return s->cams[view].path.c_str();
}
extern "C" int eg3d_sfm_image_size(const eg3d_sfm* s, int* width, int* height) {
  if (!s) return -1;
  if (width) *width = s->width;
  if (height) *height = s->height;
  return 0;
}

extern "C" int eg3d_sfm_seeds(const eg3d_sfm* s, eg3d_seeds* out) {
  if (!s || !out) return -1;
  out->n_seeds = (uint32_t)(s->trk_off.size() - 1);
  out->trk_off = s->trk_off.data();
  out->trk_view = s->trk_view.data();
  out->trk_xy = s->trk_xy.data();
  return 0;
}

extern "C" int eg3d_sfm_add_point(eg3d_sfm* s, const float* X3, int n_obs, const int32_t* views, const float* xy) {
  if (!s || n_obs < 0) return -1;
  // the structure's offsets are 32-bit (as are eg3d_gn_filter's and the observation filter's): a point that would
  // take the observation count past 2^32 - 1 is refused instead of wrapping (a cloud that large has to be added in
  // parts / filtered first)
  if ((uint64_t)s->trk_view.size() + (uint64_t)n_obs > 0xffffffffull) return -2;
  s->X.insert(s->X.end(), X3, X3 + 3);
  for (int i = 0; i < n_obs; i++) {
    s->trk_view.push_back(views[i]);
    s->trk_xy.push_back(xy[2 * i]);
    s->trk_xy.push_back(xy[2 * i + 1]);
  }
  s->trk_off.push_back((uint32_t)s->trk_view.size());
  return 0;
}

// add_3dpoints_to_sfmd (output_utilities.cpp:96-111)
extern "C" int eg3d_sfm_add_edgepoints(eg3d_sfm* s, const eg3d_edgepoints* p, const uint8_t* keep) {
  if (!s || !p) return -1;
  // all or nothing: count first, so that a cloud whose kept observations do not fit the 32-bit offsets leaves the
  // structure untouched (-2)
  uint64_t total = s->trk_view.size();
  for (uint64_t i = 0; i < p->n_points; i++) {
    if (keep && !keep[i]) continue;
    const uint64_t n = p->obs_off[i + 1] - p->obs_off[i];
    if (n > 0x7fffffffull) return -2;
    total += n;
  }
  if (total > 0xffffffffull) return -2;
  for (uint64_t i = 0; i < p->n_points; i++) {
    if (keep && !keep[i]) continue;
    const uint64_t a = p->obs_off[i], b = p->obs_off[i + 1];
    const int rc = eg3d_sfm_add_point(s, p->X + 3 * i, (int)(b - a), p->obs_view + a, p->obs_xy + 2 * (size_t)a);
    if (rc) return rc;
  }
  return 0;
}

// removeOutliers (outliers_filtering.cpp:66-92)
extern "C" int eg3d_sfm_remove_outliers(eg3d_sfm* s, const uint8_t* inlier) {
  if (!s || !inlier) return -1;
  std::vector<float> X, xy;
  std::vector<uint32_t> off{0};
  std::vector<int32_t> view;
  const size_t n = s->X.size() / 3;
  for (size_t i = 0; i < n; i++)
    if (inlier[i]) {
      X.insert(X.end(), s->X.begin() + 3 * i, s->X.begin() + 3 * i + 3);
      for (uint32_t j = s->trk_off[i]; j < s->trk_off[i + 1]; j++) {
        view.push_back(s->trk_view[j]);
        xy.push_back(s->trk_xy[2 * j]);
        xy.push_back(s->trk_xy[2 * j + 1]);
      }
      off.push_back((uint32_t)view.size());
    }
  s->X.swap(X);
  s->trk_off.swap(off);
  s->trk_view.swap(view);
  s->trk_xy.swap(xy);
  return 0;
}

extern "C" int eg3d_sfm_set_point_coords(eg3d_sfm* s, const float* X) {
  if (!s || !X) return -1;
  memcpy(s->X.data(), X, sizeof(float) * s->X.size());
  return 0;
}

extern "C" int eg3d_sfm_analytic_F(const eg3d_sfm* s, double* F, uint8_t* F_valid) {
  if (!s) return -1;
  const int V = (int)s->cams.size();
  for (int i = 0; i < V; i++)
    for (int j = 0; j < V; j++) {
      double* f = F + ((size_t)i * V + j) * 9;
      if (i == j) {
        for (int k = 0; k < 9; k++) f[k] = 0;
        F_valid[(size_t)i * V + j] = 0;
        continue;
      }
      const Cam &a = s->cams[i], &b = s->cams[j];
      eg3dh::fundamental_from_cameras(a.focal, a.ppx, a.ppy, a.R, a.t, b.focal, b.ppx, b.ppy, b.R, b.t, f);
      F_valid[(size_t)i * V + j] = 1;
    }
  return 0;
}

extern "C" int eg3d_sfm_estimate_F(const eg3d_sfm* s, int estimate, uint64_t rng_seed, double* F, uint8_t* F_valid,
                                   uint32_t* n_common) {
  if (!s) return -1;
  return eg3d_host_estimate_F((int)s->cams.size(), s->X.size() / 3, s->trk_off.data(), s->trk_view.data(),
                              s->trk_xy.data(), estimate, rng_seed, F, F_valid, n_common);
}

static bool load_json(const char* path, JVal& root) {
  std::ifstream f(path, std::ios::binary);
  if (!f) return false;
  std::stringstream ss;
  ss << f.rdbuf();
  std::string txt = ss.str();
  Parser p{txt.data(), txt.data() + txt.size()};
  root = p.parse();
  return p.ok && root.t == JVal::OBJ;
}

// Checked access for the reader: a missing key, a wrong type or a short array makes the whole
// read fail (nullptr) instead of dereferencing a null / out-of-range element.
struct Bad {};
static const JVal& need(const JVal* v, const char* k) {
  const JVal* r = (v && v->t == JVal::OBJ) ? v->get(k) : nullptr;
  if (!r) throw Bad();
  return *r;
}
static const JVal& at(const JVal& v, size_t i) {
  if (v.t != JVal::ARR || i >= v.a.size()) throw Bad();
  return v.a[i];
}
static double numv(const JVal& v) {
  if (v.t != JVal::NUM) throw Bad();
  return eg3d_json::parse_double(v.s);
}
static const std::string& strv(const JVal& v) {
  if (v.t != JVal::STR) throw Bad();
  return v.s;
}
static const JVal* data_of(const JVal& item) {  // item.value.ptr_wrapper.data, or null (entry skipped as in the reference)
  const JVal* d = item.t == JVal::OBJ ? item.get("value") : nullptr;
  d = (d && d->t == JVal::OBJ) ? d->get("ptr_wrapper") : nullptr;
  d = (d && d->t == JVal::OBJ) ? d->get("data") : nullptr;
  return (d && d->t == JVal::OBJ) ? d : nullptr;
}

static eg3d_sfm* sfm_read_json_impl(const char* path, eg3d_sfm*& s) {
  JVal root;
  if (!path || !load_json(path, root)) return nullptr;
  const JVal *views = root.get("views"), *intr = root.get("intrinsics"), *extr = root.get("extrinsics"),
             *structure = root.get("structure"), *rp = root.get("root_path");
  if (!views || !intr || !extr || !structure || views->t != JVal::ARR || intr->t != JVal::ARR || extr->t != JVal::ARR ||
      structure->t != JVal::ARR)
    return nullptr;
  s = nullptr;
  try {
    std::string base = (rp && rp->t == JVal::STR) ? rp->s : "";
    struct K {
      float f, px, py;
    };
    std::map<int, K> intrinsics;
    for (const JVal& it : intr->a) {
      const JVal* d = data_of(it);
      if (!d) continue;
      K k;
      k.f = (float)numv(need(d, "focal_length"));  // a non-pinhole intrinsic (no focal_length) fails the read
      k.px = (float)numv(at(need(d, "principal_point"), 0));
      k.py = (float)numv(at(need(d, "principal_point"), 1));
      intrinsics[(int)numv(need(&it, "key"))] = k;
    }
    struct E {
      float R[9], C[3];
    };
    std::map<int, E> extrinsics;
    std::map<int, int> map_pos;
    int pos = 0;
    for (const JVal& it : extr->a) {
      E e;
      const JVal& v = need(&it, "value");
      for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) e.R[3 * r + c] = (float)numv(at(at(need(&v, "rotation"), r), c));
      for (int r = 0; r < 3; r++) e.C[r] = (float)numv(at(need(&v, "center"), r));
      const int key = (int)numv(need(&it, "key"));
      extrinsics[key] = e;
      map_pos[key] = pos++;
    }
    s = new eg3d_sfm();
    s->cams.resize(views->a.size());
    for (size_t i = 0; i < views->a.size(); i++) {
      const JVal* d = data_of(views->a[i]);
      if (!d) continue;
      Cam& c = s->cams[i];
      c.path = base + strv(need(d, "local_path")) + strv(need(d, "filename"));
      s->width = (int)numv(need(d, "width"));
      s->height = (int)numv(need(d, "height"));
      const int ii = (int)numv(need(d, "id_intrinsic")), ie = (int)numv(need(d, "id_pose"));
      auto ki = intrinsics.find(ii);
      auto ke = extrinsics.find(ie);
      if (ki == intrinsics.end() || ke == extrinsics.end()) continue;  // reference prints and leaves zeros
      c.focal = ki->second.f;
      c.ppx = ki->second.px;
      c.ppy = ki->second.py;
      memcpy(c.R, ke->second.R, sizeof(c.R));
      memcpy(c.C, ke->second.C, sizeof(c.C));
      eg3dh::translation_from_center(c.R, c.C, c.t);
      eg3dh::camera_matrix(c.focal, c.ppx, c.ppy, c.R, c.t, c.P);
    }
    s->refresh_P();
    for (const JVal& pt : structure->a) {
      const JVal& v = need(&pt, "value");
      const JVal& Xj = need(&v, "X");
      float X[3] = {(float)numv(at(Xj, 0)), (float)numv(at(Xj, 1)), (float)numv(at(Xj, 2))};
      std::vector<int32_t> vw;
      std::vector<float> xy;
      const JVal& obs = need(&v, "observations");
      if (obs.t != JVal::ARR) throw Bad();
      for (const JVal& ob : obs.a) {
        auto it = map_pos.find((int)numv(need(&ob, "key")));
        if (it == map_pos.end()) continue;
        vw.push_back(it->second);
        const JVal& x = need(&need(&ob, "value"), "x");
        xy.push_back((float)numv(at(x, 0)));
        xy.push_back((float)numv(at(x, 1)));
      }
      eg3d_sfm_add_point(s, X, (int)vw.size(), vw.data(), xy.data());
    }
  } catch (const Bad&) {
    delete s;
    return nullptr;
  }
  return s;
}

// The camera model of the reader / eg3d_sfm_set_camera on bare arrays (include/eg3d_host.h): tests pin it against
// the reference's vendored glm (tests/test_glm_pin.py)
extern "C" int eg3d_host_camera_model(uint64_t n, const float* fpp, const float* R9, const float* C3, float* t3, float* P16) {
  if (!fpp || !R9 || !C3 || !t3 || !P16) return -1;
  for (uint64_t i = 0; i < n; i++) {
    eg3dh::translation_from_center(R9 + 9 * i, C3 + 3 * i, t3 + 3 * i);
    eg3dh::camera_matrix(fpp[3 * i], fpp[3 * i + 1], fpp[3 * i + 2], R9 + 9 * i, t3 + 3 * i, P16 + 16 * i);
  }
  return 0;
}

// Diagnostics of the writer's text rules (include/eg3d_host.h), used by tests/test_json_rapidjson.py
extern "C" int eg3d_host_json_double_text(double d, char* buf, int cap) {
  if (!buf || cap < 2 || !(d == d) || d > 1.7976931348623157e308 || d < -1.7976931348623157e308) return -1;
  const std::string s = eg3d_json::double_text(d);
  if ((int)s.size() + 1 > cap) return -1;
  memcpy(buf, s.c_str(), s.size() + 1);
  return (int)s.size();
}
extern "C" int eg3d_host_json_number_text(const char* literal, char* buf, int cap) {
  if (!literal || !buf || cap < 2) return -1;
  bool ok = true;
  const std::string s = eg3d_json::normalize_number(literal, &ok);
  if (!ok || (int)s.size() + 1 > cap) return -1;
  memcpy(buf, s.c_str(), s.size() + 1);
  return (int)s.size();
}
extern "C" int eg3d_host_json_cached_power(int index, uint64_t* f, int* e) {
  if (index < 0 || index > 86 || !f || !e) return -1;
  const eg3d_json::Fp p = eg3d_json::cached_powers()[index];
  *f = p.f;
  *e = p.e;
  return 0;
}

// No C++ exception crosses the C ABI: a file too large for memory (std::bad_alloc from the DOM or the point arrays)
// is a failed read, not std::terminate.
extern "C" eg3d_sfm* eg3d_sfm_read_json(const char* path) {
  eg3d_sfm* s = nullptr;
  try {
    return sfm_read_json_impl(path, s);
  } catch (...) {
    delete s;
    return nullptr;
  }
}

static int sfm_write_json_impl(const eg3d_sfm* s, const char* in_path, const char* out_path);
// The text write_val prints for points [i0, i1) of the "structure" array (elements at indent level 2, 4 spaces per level),
// without building the document: "key" = the point's index, "X" = its coordinates, one observation object per track entry
// ("key" = view, "value": {"id_feat": 0, "x": [x, y]}). Numbers are the writer's text of the float widened to double (the
// document path prints normalize_number(double_text(d)), which is the identity on double_text's output: Grisu text reads
// back to the same double — tests/test_host_io.py checks it on random floats).
static void structure_text(const eg3d_sfm* s, size_t i0, size_t i1, size_t n_total, std::string& out, bool& nonfinite) {
  out.clear();
  out.reserve((i1 - i0) * 2048);
  auto ind = [&](int n) { out.append((size_t)n * 4, ' '); };
  auto num = [&](float v) {
    const double d = (double)v;
    if (!(d == d) || d > 1.7e308 || d < -1.7e308) {
      nonfinite = true;
      out += "null";
    } else {
      out += eg3d_json::double_text(d);
    }
  };
  auto integer = [&](long long v) {
    char buf[24];
    const int k = snprintf(buf, sizeof(buf), "%lld", v);
    out.append(buf, (size_t)k);
  };
  for (size_t i = i0; i < i1; i++) {
    ind(2); out += "{\n";
    ind(3); out += "\"key\": "; integer((long long)i); out += ",\n";
    ind(3); out += "\"value\": {\n";
    ind(4); out += "\"X\": [\n";
    for (int k = 0; k < 3; k++) {
      ind(5); num(s->X[3 * i + k]); out += k < 2 ? ",\n" : "\n";
    }
    ind(4); out += "],\n";
    ind(4); out += "\"observations\": ";
    const uint32_t a = s->trk_off[i], b = s->trk_off[i + 1];
    if (a == b) {
      out += "[]\n";
    } else {
      out += "[\n";
      for (uint32_t j = a; j < b; j++) {
        ind(5); out += "{\n";
        ind(6); out += "\"key\": "; integer(s->trk_view[j]); out += ",\n";
        ind(6); out += "\"value\": {\n";
        ind(7); out += "\"id_feat\": 0,\n";
        ind(7); out += "\"x\": [\n";
        ind(8); num(s->trk_xy[2 * j]); out += ",\n";
        ind(8); num(s->trk_xy[2 * j + 1]); out += "\n";
        ind(7); out += "]\n";
        ind(6); out += "}\n";
        ind(5); out += j + 1 < b ? "},\n" : "}\n";
      }
      ind(4); out += "]\n";
    }
    ind(3); out += "}\n";
    ind(2); out += i + 1 < n_total ? "},\n" : "}\n";
  }
}

extern "C" int eg3d_sfm_write_json(const eg3d_sfm* s, const char* in_path, const char* out_path) {
  try {
    return sfm_write_json_impl(s, in_path, out_path);
  } catch (...) {
    return -1;
  }
}

static int sfm_write_json_impl(const eg3d_sfm* s, const char* in_path, const char* out_path) {
  if (!s || !out_path) return -1;
  g_nonfinite_written = false;
  JVal in;
  bool have_in = in_path && load_json(in_path, in);
  JVal root;
  root.t = JVal::OBJ;
  auto copy_or = [&](const char* k, JVal def) {
    const JVal* v = have_in ? in.get(k) : nullptr;
    root.o.emplace_back(k, v ? *v : def);
  };
  JVal empty_arr;
  empty_arr.t = JVal::ARR;
  JVal empty_str;
  empty_str.t = JVal::STR;
  copy_or("sfm_data_version", empty_str);
  copy_or("root_path", empty_str);
  if (have_in && in.get("views") && in.get("intrinsics")) {
    copy_or("views", empty_arr);
    copy_or("intrinsics", empty_arr);
  } else {
    // no input file to pass through (a scene built with eg3d_sfm_create): write the views and one
    // pinhole intrinsic per camera in the layout OpenMvgParser reads (OpenMvgParser.cpp:39-301)
    auto jstr = [](const std::string& v) {
      JVal j;
      j.t = JVal::STR;
      j.s = v;
      return j;
    };
    JVal views, intr;
    views.t = intr.t = JVal::ARR;
    for (size_t i = 0; i < s->cams.size(); i++) {
      const Cam& c = s->cams[i];
      JVal v, val, pw, d;
      v.t = val.t = pw.t = d.t = JVal::OBJ;
      d.o.emplace_back("local_path", jstr("/"));
      d.o.emplace_back("filename", jstr(c.path));
      d.o.emplace_back("width", jint(s->width));
      d.o.emplace_back("height", jint(s->height));
      d.o.emplace_back("id_view", jint((long long)i));
      d.o.emplace_back("id_intrinsic", jint((long long)i));
      d.o.emplace_back("id_pose", jint((long long)i));
      pw.o.emplace_back("id", jint(2147483649ll + (long long)i));
      pw.o.emplace_back("data", d);
      val.o.emplace_back("polymorphic_id", jint(1073741824ll));
      val.o.emplace_back("ptr_wrapper", pw);
      v.o.emplace_back("key", jint((long long)i));
      v.o.emplace_back("value", val);
      views.a.push_back(v);
      JVal k, kval, kpw, kd, pp;
      k.t = kval.t = kpw.t = kd.t = JVal::OBJ;
      pp.t = JVal::ARR;
      pp.a.push_back(jnum(c.ppx));
      pp.a.push_back(jnum(c.ppy));
      kd.o.emplace_back("width", jint(s->width));
      kd.o.emplace_back("height", jint(s->height));
      kd.o.emplace_back("focal_length", jnum(c.focal));
      kd.o.emplace_back("principal_point", pp);
      kpw.o.emplace_back("id", jint(2147484649ll + (long long)i));
      kpw.o.emplace_back("data", kd);
      kval.o.emplace_back("polymorphic_id", jint(2147483649ll));
      kval.o.emplace_back("polymorphic_name", jstr("pinhole"));
      kval.o.emplace_back("ptr_wrapper", kpw);
      k.o.emplace_back("key", jint((long long)i));
      k.o.emplace_back("value", kval);
      intr.a.push_back(k);
    }
    root.o.emplace_back("views", views);
    root.o.emplace_back("intrinsics", intr);
  }
  JVal ex;
  ex.t = JVal::ARR;
  for (size_t i = 0; i < s->cams.size(); i++) {
    const Cam& c = s->cams[i];
    JVal e, pose, rot, cen;
    e.t = pose.t = JVal::OBJ;
    rot.t = cen.t = JVal::ARR;
    for (int r = 0; r < 3; r++) {
      JVal row;
      row.t = JVal::ARR;
      for (int cc = 0; cc < 3; cc++) row.a.push_back(jnum(c.R[3 * r + cc]));
      rot.a.push_back(row);
      cen.a.push_back(jnum(c.C[r]));
    }
    pose.o.emplace_back("rotation", rot);
    pose.o.emplace_back("center", cen);
    e.o.emplace_back("key", jint((long long)i));
    e.o.emplace_back("value", pose);
    ex.a.push_back(e);
  }
  root.o.emplace_back("extrinsics", ex);
  copy_or("control_points", empty_arr);
  // ---- the file. Everything but "structure" is small and goes through the document writer; the structure — one entry per
  // point, 5 KB of text each with its observations: 1.7 GB for a dtu006-sized cloud — is STREAMED: the text write_val
  // would print for it is generated straight from the arrays, chunk by chunk on the host's cores, and written in order.
  // (Until round 6 the structure was built as a document tree first — a std::string per number, a deep copy per
  // push_back — and every number went text -> strtod -> text again on its way out: 15 s for that file, 160x the GPU's
  // share of the run. Same bytes: tests/test_json_rapidjson.py, tests/test_host_io.py.)
  std::ofstream f(out_path, std::ios::binary);
  if (!f) return -1;
  f << "{\n";
  const JVal control_points = root.o.back().second;
  root.o.pop_back();
  for (auto& kv : root.o) {
    indent(f, 1);
    f << '"' << eg3d_json::escape_string(kv.first) << "\": ";
    write_val(f, kv.second, 1);
    f << ",\n";
  }
  indent(f, 1);
  f << "\"structure\": ";
  const size_t n = s->X.size() / 3;
  bool nonfinite = false;
  if (!n) {
    f << "[]";
  } else {
    f << "[\n";
    const size_t CH = 256;  // points per chunk
    const size_t n_chunks = (n + CH - 1) / CH;
    int threads = 1;
#ifdef _OPENMP
    threads = std::max(1, std::min(omp_get_max_threads(), 32));
#endif
    const size_t wave = (size_t)threads * 2;
    std::vector<std::string> text(wave);
    std::vector<char> bad(wave, 0);
    for (size_t c0 = 0; c0 < n_chunks; c0 += wave) {
      const size_t c1 = std::min(n_chunks, c0 + wave);
#pragma omp parallel for schedule(dynamic, 1)
      for (long long c = (long long)c0; c < (long long)c1; c++) {
        bool nf = false;
        structure_text(s, (size_t)c * CH, std::min(n, ((size_t)c + 1) * CH), n, text[(size_t)c - c0], nf);
        bad[(size_t)c - c0] = nf ? 1 : 0;
      }
      for (size_t c = c0; c < c1; c++) {
        f.write(text[c - c0].data(), (std::streamsize)text[c - c0].size());
        nonfinite = nonfinite || bad[c - c0];
      }
      if (!f.good()) return -1;
    }
    indent(f, 1);
    f << "]";
  }
  f << ",\n";
  indent(f, 1);
  f << "\"control_points\": ";
  write_val(f, control_points, 1);
  f << "\n}";
  if (g_nonfinite_written || nonfinite) return -2;  // a NaN / infinite coordinate was written as null: not a valid SfM file
  return f.good() ? 0 : -1;
}
