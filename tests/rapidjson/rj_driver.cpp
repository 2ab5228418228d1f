// rj_driver.cpp — TEST-ONLY driver over the rapidjson that the reference tree vendors (header-only,
// /root/reference/external/rapidjson; compiled in the build container only, nothing of it is copied here).
// It produces what the REFERENCE's JSON layer produces, so that the product's own reader / writer
// (edgegraph3d_amd/host/sfm_json.cpp, json_text.hpp) can be compared with it byte for byte:
//   doubles <in.bin> <out.txt>     every double of the file as Writer<> prints it, one per line
//   literals <in.txt> <out.txt>    every line = a JSON number literal: parsed with the default flags (as
//                                  OpenMvgParser.cpp:49-50 / output_sfm_data.cpp:187-193 do), re-printed by Writer<>
//   powers <out.txt>               Grisu's cached powers table: "f e" per index 0..86
//   rewrite <in.json> <sfm.bin> <out.json>
//                                  the document output_sfm_data writes (output_sfm_data.cpp:186-229): version,
//                                  root_path, views, intrinsics, control_points copied from <in.json>; extrinsics and
//                                  structure built from the floats of <sfm.bin> as Value(float) / Value(int), printed
//                                  with PrettyWriter<OStreamWrapper>
//   index <in.json> <out.txt>      the fields OpenMvgParser.cpp:39-301 reads, with the accessors it uses
//                                  (GetInt / GetFloat / GetDouble), as a flat text dump
#include <rapidjson/document.h>
#include <rapidjson/internal/dtoa.h>
#include <rapidjson/istreamwrapper.h>
#include <rapidjson/ostreamwrapper.h>
#include <rapidjson/prettywriter.h>
#include <rapidjson/stringbuffer.h>
#include <rapidjson/writer.h>

#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <map>
#include <string>
#include <vector>

using namespace rapidjson;

static std::vector<char> slurp(const char* p) {
  std::ifstream f(p, std::ios::binary);
  return std::vector<char>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  const std::string mode = argv[1];
  if (mode == "doubles" && argc == 4) {
    std::vector<char> b = slurp(argv[2]);
    std::ofstream o(argv[3]);
    for (size_t i = 0; i + 8 <= b.size(); i += 8) {
      double d;
      memcpy(&d, &b[i], 8);
      StringBuffer sb;
      Writer<StringBuffer> w(sb);
      w.Double(d);
      o << sb.GetString() << "\n";
    }
    return 0;
  }
  if (mode == "literals" && argc == 4) {
    std::ifstream in(argv[2]);
    std::ofstream o(argv[3]);
    std::string line;
    while (std::getline(in, line)) {
      Document d;
      d.Parse(line.c_str());
      if (d.HasParseError()) {
        o << "ERROR\n";
        continue;
      }
      StringBuffer sb;
      Writer<StringBuffer> w(sb);
      d.Accept(w);
      o << sb.GetString() << "\n";
    }
    return 0;
  }
  if (mode == "powers" && argc == 3) {
    std::ofstream o(argv[2]);
    for (size_t i = 0; i < 87; i++) {
      const internal::DiyFp p = internal::GetCachedPowerByIndex(i);
      o << p.f << " " << p.e << "\n";
    }
    return 0;
  }
  if (mode == "rewrite" && argc == 5) {
    std::ifstream ifs(argv[2]);
    IStreamWrapper isw(ifs);
    Document in;
    in.ParseStream(isw);
    if (!in.IsObject()) return 3;
    std::vector<char> b = slurp(argv[3]);
    const char* p = b.data();
    auto rd_i = [&]() { int32_t v; memcpy(&v, p, 4); p += 4; return v; };
    auto rd_f = [&]() { float v; memcpy(&v, p, 4); p += 4; return v; };
    Document doc;
    doc.SetObject();
    Document::AllocatorType& a = doc.GetAllocator();
    Value root(kObjectType);
    root.AddMember("sfm_data_version", Value(in["sfm_data_version"], a), a);
    root.AddMember("root_path", Value(in["root_path"], a), a);
    root.AddMember("views", Value(in["views"], a), a);
    root.AddMember("intrinsics", Value(in["intrinsics"], a), a);
    const int V = rd_i();
    Value ex(kArrayType);
    for (int i = 0; i < V; i++) {
      Value rot(kArrayType);
      for (int r = 0; r < 3; r++) {
        Value row(kArrayType);
        for (int c = 0; c < 3; c++) row.PushBack(Value(rd_f()), a);
        rot.PushBack(row, a);
      }
      Value cen(kArrayType);
      for (int r = 0; r < 3; r++) cen.PushBack(Value(rd_f()), a);
      Value pose(kObjectType);
      pose.AddMember("rotation", rot, a);
      pose.AddMember("center", cen, a);
      Value e(kObjectType);
      e.AddMember("key", i, a);
      e.AddMember("value", pose, a);
      ex.PushBack(e, a);
    }
    root.AddMember("extrinsics", ex, a);
    const int N = rd_i();
    Value st(kArrayType);
    for (int i = 0; i < N; i++) {
      Value X(kArrayType);
      for (int k = 0; k < 3; k++) X.PushBack(Value(rd_f()), a);
      const int no = rd_i();
      Value obs(kArrayType);
      for (int j = 0; j < no; j++) {
        const int cam = rd_i();
        Value x(kArrayType);
        x.PushBack(Value(rd_f()), a);
        x.PushBack(Value(rd_f()), a);
        Value ov(kObjectType);
        ov.AddMember("id_feat", 0, a);
        ov.AddMember("x", x, a);
        Value o(kObjectType);
        o.AddMember("key", cam, a);
        o.AddMember("value", ov, a);
        obs.PushBack(o, a);
      }
      Value val(kObjectType);
      val.AddMember("X", X, a);
      val.AddMember("observations", obs, a);
      Value pt(kObjectType);
      pt.AddMember("key", (uint64_t)i, a);
      pt.AddMember("value", val, a);
      st.PushBack(pt, a);
    }
    root.AddMember("structure", st, a);
    root.AddMember("control_points", Value(in["control_points"], a), a);
    std::ofstream ofs(argv[4]);
    OStreamWrapper osw(ofs);
    PrettyWriter<OStreamWrapper> w(osw);
    root.Accept(w);
    return 0;
  }
  if (mode == "index" && argc == 4) {
    std::ifstream ifs(argv[2]);
    IStreamWrapper isw(ifs);
    Document d;
    d.ParseStream(isw);
    if (!d.IsObject()) return 3;
    std::ofstream o(argv[3]);
    char buf[64];
    auto hexf = [&](float v) { uint32_t u; memcpy(&u, &v, 4); snprintf(buf, sizeof(buf), "%08x", u); return std::string(buf); };
    const Value& K = d["intrinsics"];
    for (SizeType i = 0; i < K.Size(); i++) {
      const Value& dd = K[i]["value"]["ptr_wrapper"]["data"];
      o << "K " << K[i]["key"].GetInt() << " " << hexf(dd["focal_length"].GetFloat()) << " "
        << hexf(dd["principal_point"][0].GetFloat()) << " " << hexf(dd["principal_point"][1].GetFloat()) << "\n";
    }
    const Value& E = d["extrinsics"];
    std::map<int, int> map_pos;
    for (SizeType i = 0; i < E.Size(); i++) {
      map_pos[E[i]["key"].GetInt()] = (int)i;
      o << "E " << E[i]["key"].GetInt();
      for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) o << " " << hexf(E[i]["value"]["rotation"][r][c].GetFloat());
      for (int r = 0; r < 3; r++) o << " " << hexf(E[i]["value"]["center"][r].GetFloat());
      o << "\n";
    }
    const Value& Vw = d["views"];
    for (SizeType i = 0; i < Vw.Size(); i++) {
      const Value& dd = Vw[i]["value"]["ptr_wrapper"]["data"];
      o << "V " << dd["width"].GetInt() << " " << dd["height"].GetInt() << " " << dd["id_intrinsic"].GetInt() << " "
        << dd["id_pose"].GetInt() << " " << d["root_path"].GetString() << dd["local_path"].GetString()
        << dd["filename"].GetString() << "\n";
    }
    const Value& S = d["structure"];
    for (SizeType i = 0; i < S.Size(); i++) {
      const Value& X = S[i]["value"]["X"];
      o << "P " << hexf(X[0].GetFloat()) << " " << hexf(X[1].GetFloat()) << " " << hexf(X[2].GetFloat());
      const Value& ob = S[i]["value"]["observations"];
      for (SizeType j = 0; j < ob.Size(); j++) {
        const Value& x = ob[j]["value"]["x"];
        o << " " << map_pos.at(ob[j]["key"].GetInt()) << ":" << hexf((float)x[0].GetDouble()) << ":" << hexf((float)x[1].GetDouble());
      }
      o << "\n";
    }
    return 0;
  }
  return 2;
}
