// A call logger for the class-method stream (test infrastructure of the mirror).
//
// The mirror exists because GHC does not in this image; what makes it worth testing is that it issues the calls
// `instance Tensor HipT` (hs/TensorOps/Backend/HipTensor.hs) would issue for the same TOp.  With the logger on, every
// `HipT::` method appends one line -- method, static arguments, the identities of its operands and of its result,
// the result's shape -- and tests/test_call_trace.py compares the logged dataflow graph with the one the oracle's
// restatement of the DSL (oracle/top.py on a tracing backend) produces for the same program.
//
// Line format (tab separated):  method  params(,)  input ids(,)  output id  dims(x)|batch
// `L id dims|batch` introduces a handle that no logged call produced and the caller did not name.
#pragma once
#include <cstdio>
#include <map>
#include <string>
#include <vector>

#include "../../../include/tensorops_hip.h"

namespace tensorops {
namespace trace {

struct State {
  bool on = false;
  std::map<to_tensor, int> ids;
  std::vector<to_tensor> keep;  // retained: an address must not come back as another value while the log is open
  std::string log;
  int next = 0;
};
inline State& st() {
  static thread_local State s;
  return s;
}
inline bool on() { return st().on; }

inline std::string shape_of(to_tensor h) {
  int rank = 0;
  int64_t d[TO_MAX_RANK], b = 0;
  to_shape(h, &rank, d, &b);
  std::string s;
  for (int i = 0; i < rank; ++i) s += (i ? "x" : "") + std::to_string(d[i]);
  return s + "|" + std::to_string(b);
}

inline int name(to_tensor h, int id) {
  State& s = st();
  s.ids[h] = id;
  to_retain(h);
  s.keep.push_back(h);
  return id;
}

inline int id_of(to_tensor h) {
  State& s = st();
  auto it = s.ids.find(h);
  if (it != s.ids.end()) return it->second;
  const int id = name(h, s.next++);
  s.log += "L\t" + std::to_string(id) + "\t" + shape_of(h) + "\n";
  return id;
}

inline void begin(int n, const to_tensor* leaves) {
  State& s = st();
  for (to_tensor h : s.keep) to_release(h);
  s = State();
  s.on = true;
  for (int i = 0; i < n; ++i)
    if (!s.ids.count(leaves[i])) name(leaves[i], i);
  s.next = n;
}

inline std::string end() {
  State& s = st();
  s.on = false;
  for (to_tensor h : s.keep) to_release(h);
  s.keep.clear();
  s.ids.clear();
  std::string out;
  out.swap(s.log);
  return out;
}

inline std::string num(double v) {
  char b[40];
  std::snprintf(b, sizeof b, "%.6e", v);
  return b;
}

inline void call(const char* method, const std::vector<std::string>& params, const std::vector<to_tensor>& ins,
                 to_tensor out) {
  if (!on()) return;
  State& s = st();
  std::string line = method;
  line += "\t";
  for (size_t i = 0; i < params.size(); ++i) line += (i ? "," : "") + params[i];
  line += "\t";
  std::vector<int> in_ids;
  for (to_tensor h : ins) in_ids.push_back(id_of(h));  // (may append L lines first)
  for (size_t i = 0; i < in_ids.size(); ++i) line += (i ? "," : "") + std::to_string(in_ids[i]);
  int oid;
  auto it = s.ids.find(out);
  if (it != s.ids.end()) oid = it->second;  // the same value again (a memo hit, `sumT [x] = x`)
  else oid = name(out, s.next++);
  line += "\t" + std::to_string(oid) + "\t" + shape_of(out) + "\n";
  s.log += line;
}

}  // namespace trace
}  // namespace tensorops
