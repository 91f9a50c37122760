// tests/native/ckpt_tool.cpp -- the checkpoint container of ga3c_net_save / ga3c_net_load (ga3c_amd/csrc/ga3c_checkpoint.hpp)
// on its own, so that it can be held to numpy without a GPU (tests/test_checkpoint_container_cpu.py).
//   ckpt_tool write <out.npz>   three members: "w/x:0" f32 [2,3] = 0..5, "b:0" f32 [4] = 0.5 k, "step" int64 () = 1234567890123
//   ckpt_tool read <in.npz>     one line per member: name|descr|shape,..|first values as f32 or i64
#include <cstdio>
#include <cstring>

#include "../../ga3c_amd/csrc/ga3c_checkpoint.hpp"

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  std::string err;
  if (!strcmp(argv[1], "write")) {
    std::vector<ga3c_ckpt::Member> ms(3);
    float w[6] = {0, 1, 2, 3, 4, 5}, b[4] = {0.f, 0.5f, 1.f, 1.5f};
    const int64_t step = 1234567890123LL;
    ms[0].name = "w/x:0"; ms[0].descr = "<f4"; ms[0].shape = {2, 3};
    ms[0].bytes.assign((uint8_t*)w, (uint8_t*)w + sizeof w);
    ms[1].name = "b:0"; ms[1].descr = "<f4"; ms[1].shape = {4};
    ms[1].bytes.assign((uint8_t*)b, (uint8_t*)b + sizeof b);
    ms[2].name = "step"; ms[2].descr = "<i8";
    ms[2].bytes.assign((const uint8_t*)&step, (const uint8_t*)&step + 8);
    if (!ga3c_ckpt::write_npz(argv[2], ms, &err)) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
    return 0;
  }
  std::map<std::string, ga3c_ckpt::Member> ms;
  if (!ga3c_ckpt::read_npz(argv[2], &ms, &err)) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
  for (auto& kv : ms) {
    const ga3c_ckpt::Member& m = kv.second;
    printf("%s|%s|", m.name.c_str(), m.descr.c_str());
    for (size_t i = 0; i < m.shape.size(); ++i) printf("%s%lld", i ? "," : "", (long long)m.shape[i]);
    printf("|");
    if (m.descr == "<i8") { int64_t v; memcpy(&v, m.bytes.data(), 8); printf("%lld", (long long)v); }
    else for (size_t i = 0; i < m.bytes.size() / 4 && i < 8; ++i) { float v; memcpy(&v, &m.bytes[4 * i], 4); printf("%s%.9g", i ? "," : "", v); }
    printf("\n");
  }
  return 0;
}
