// cpslab_host.cpp -- see cpslab_host.hpp.  Pure host code over the C ABI (include/apss.h); links libapss_hip.so.
#include "cpslab_host.hpp"

#include <fstream>
#include <iterator>
#include <map>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <sstream>

#include "../../include/apss.h"

namespace cpslab {

SparseVector::SparseVector(int size_, std::vector<int32_t> idx, std::vector<double> val)
    : size(size_), indices(std::move(idx)), values(std::move(val)) {
  if (indices.size() != values.size()) throw std::invalid_argument("requirement failed");  // SparseVector.scala:202
}

std::string SparseVector::toString() const {
  std::ostringstream o;
  o.precision(17);
  o << "(" << size << ",[";
  for (size_t i = 0; i < indices.size(); ++i) o << (i ? "," : "") << indices[i];
  o << "],[";
  for (size_t i = 0; i < values.size(); ++i) o << (i ? "," : "") << values[i];
  o << "])";
  return o.str();
}

SparseVector SparseVector::fromString(const std::string &s) {
  // inputString.split(",\\[") must give 3 pieces (SparseVector.scala:133-136)
  const size_t p1 = s.find(",[");
  const size_t p2 = p1 == std::string::npos ? p1 : s.find(",[", p1 + 2);
  if (p1 == std::string::npos || p2 == std::string::npos || s.find(",[", p2 + 2) != std::string::npos)
    throw std::runtime_error("cannot parse " + s);
  auto strip = [](std::string t, const char *drop) {
    for (const char *d = drop; *d; ++d) t.erase(std::remove(t.begin(), t.end(), *d), t.end());
    return t;
  };
  SparseVector v;
  v.size = std::stoi(strip(s.substr(0, p1), "("));
  std::stringstream is(strip(s.substr(p1 + 2, p2 - p1 - 2), "]")), vs(strip(s.substr(p2 + 2), "])"));
  std::string tok;
  while (std::getline(is, tok, ',')) if (!tok.empty()) v.indices.push_back(std::stoi(tok));
  while (std::getline(vs, tok, ',')) if (!tok.empty()) v.values.push_back(std::stod(tok));
  if (v.indices.size() != v.values.size()) throw std::invalid_argument("requirement failed");
  return v;
}

std::pair<std::string, SparkSparseVector> CCWEBVideoLoadGenerator::lineParser(const std::string &line) {
  // line.replace("(", "").replace(")", "").replace("[", "").replace("]", "").split(",")  (CCW:11-12); String.split drops
  // trailing empty strings
  std::string t;
  for (char ch : line)
    if (ch != '(' && ch != ')' && ch != '[' && ch != ']') t.push_back(ch);
  std::vector<std::string> f;
  size_t b = 0;
  for (;;) {
    const size_t e = t.find(',', b);
    f.push_back(t.substr(b, e == std::string::npos ? e : e - b));
    if (e == std::string::npos) break;
    b = e + 1;
  }
  while (!f.empty() && f.back().empty()) f.pop_back();
  if (f.size() < 2) throw std::invalid_argument("CC_WEB_VIDEO line without id and size: " + line);
  auto to_int = [&](const std::string &s) {
    size_t used = 0;
    int v = 0;
    try { v = std::stoi(s, &used); } catch (const std::exception &) { used = std::string::npos; }
    if (used != s.size() || s.empty()) throw std::invalid_argument("not an Int: '" + s + "'");
    return v;
  };
  auto to_double = [&](const std::string &s) {
    size_t used = 0;
    double v = 0;
    try { v = std::stod(s, &used); } catch (const std::exception &) { used = std::string::npos; }
    if (used != s.size() || s.empty()) throw std::invalid_argument("not a Double: '" + s + "'");
    return v;
  };
  const int size = to_int(f[1]);  // propertyArray(1).toInt
  if (size < 0) throw std::invalid_argument("negative vector size");
  // propertyArray.takeRight(vectorSize): the last `size` fields (all of them when there are fewer -- allValues(i) then
  // fails for i past the end, CCW:17)
  if ((size_t)size > f.size()) throw std::invalid_argument("fewer than `size` values: " + line);
  std::vector<int32_t> idx;
  std::vector<double> val;
  for (int i = 0; i < size; ++i) {
    const double v = to_double(f[f.size() - (size_t)size + (size_t)i]);
    if (v != 0) {  // allValues(_) != 0
      idx.push_back(i);
      val.push_back(v);
    }
  }
  return {f[0], SparseVector(size, std::move(idx), std::move(val))};
}

std::vector<std::pair<std::string, SparkSparseVector>> CCWEBVideoLoadGenerator::generateVectors() const {
  std::ifstream in(path_);
  if (!in) throw std::runtime_error("cannot open " + path_);  // Source.fromFile: FileNotFoundException
  std::vector<std::pair<std::string, SparkSparseVector>> out;
  std::string line;
  while (std::getline(in, line)) {
    if (!line.empty() && line.back() == '\r') line.pop_back();
    out.push_back(lineParser(line));
  }
  return out;
}

namespace etl {
int32_t javaStringHashCode(const std::string &s) {
  uint32_t h = 0;
  for (unsigned char ch : s) h = 31u * h + ch;
  return (int32_t)h;
}

int32_t nonNegativeMod(int32_t x, int32_t mod) {
  const int32_t r = x % mod;  // truncated, like the JVM's %
  return r < 0 ? r + mod : r;
}

std::vector<std::string> documentTokens(const std::string &path) {
  std::ifstream in(path, std::ios::binary);
  if (!in) throw std::runtime_error("cannot open " + path);
  std::string raw((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
  // BufferedReader.readLine: "\n", "\r" and "\r\n" end a line; a final terminator does not start another line
  std::string s;
  size_t i = 0;
  while (i < raw.size()) {
    size_t j = i;
    while (j < raw.size() && raw[j] != '\n' && raw[j] != '\r') ++j;
    s.append(raw, i, j - i);
    s.push_back(' ');
    if (j < raw.size() && raw[j] == '\r' && j + 1 < raw.size() && raw[j + 1] == '\n') ++j;
    i = j + 1;
  }
  s += "null ";
  std::vector<std::string> toks;
  size_t b = 0;
  for (;;) {
    const size_t e = s.find(' ', b);
    toks.push_back(s.substr(b, e == std::string::npos ? e : e - b));
    if (e == std::string::npos) break;
    b = e + 1;
  }
  while (!toks.empty() && toks.back().empty()) toks.pop_back();  // String.split drops trailing empty strings
  return toks;
}

SparseVector hashingTF(const std::vector<std::string> &tokens, int32_t numFeatures) {
  std::map<int32_t, double> tf;
  for (const auto &t : tokens) tf[nonNegativeMod(javaStringHashCode(t), numFeatures)] += 1.0;
  std::vector<int32_t> idx;
  std::vector<double> val;
  for (const auto &kv : tf) {
    idx.push_back(kv.first);
    val.push_back(kv.second);
  }
  return SparseVector(numFeatures, std::move(idx), std::move(val));
}

std::vector<SparseVector> tfidf(const std::vector<SparseVector> &tf, bool normalize) {
  std::map<int32_t, int64_t> df;
  for (const auto &v : tf)
    for (int32_t i : v.indices) ++df[i];
  const double m = (double)tf.size();
  std::vector<SparseVector> out;
  for (const auto &v : tf) {
    std::vector<double> w(v.values.size());
    double ss = 0;
    for (size_t k = 0; k < w.size(); ++k) {
      w[k] = v.values[k] * std::log((m + 1.0) / ((double)df[v.indices[k]] + 1.0));
      ss += w[k] * w[k];
    }
    if (normalize && ss > 0) {
      const double nrm = std::sqrt(ss);
      for (double &x : w) x /= nrm;
    }
    out.emplace_back(v.size, v.indices, std::move(w));
  }
  return out;
}
}  // namespace etl

std::string SimilarityOutput::toString() const {  // Message.scala:23-34
  std::ostringstream o;
  for (const auto &q : output) {
    o << "---------------------------------" << q.first << ":";
    for (const auto &c : q.second) o << c.first << "," << c.second << ";";
    o << "\n";
  }
  return o.str();
}

static int64_t now_ms() {
  return std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::system_clock::now().time_since_epoch()).count();
}

GpuIndexingWorker::GpuIndexingWorker(const Config &conf, ReplyTo replyTo) : conf_(conf), reply_to_(std::move(replyTo)) {
  apss_config c{};
  c.struct_size = (int32_t)sizeof(c);
  c.dim = conf.vectorDim;
  c.theta = conf.similarityThreshold;
  c.index_threshold = conf.indexThreshold;
  c.flags = conf.applyIndexThreshold ? APSS_FLAG_VALUE_PRUNE : 0u;
  c.device_id = conf.deviceId;
  c.tile_rows = conf.tileRows;
  c.head_terms = conf.headTerms;
  if (conf.devices.size() > 1 || conf.groupFlags) {
    // the term-sharded index of the node behind one object: the reference's DataPacket fan-out to its term workers
    // (WriteWorkerActor.scala:164-183, EntryProxyActor.scala:37-49) happens inside the library
    std::vector<int32_t> dev(conf.devices.begin(), conf.devices.end());
    if (dev.empty()) dev.push_back(conf.deviceId);
    const int32_t rc = apss_group_create(&c, (int32_t)dev.size(), dev.data(), conf.groupFlags, &g_);
    if (rc != APSS_OK) throw std::runtime_error(std::string("apss_group_create: ") + apss_group_last_error(nullptr));
    return;
  }
  if (conf.devices.size() == 1) c.device_id = conf.devices[0];
  const int32_t rc = apss_create(&c, &h_);
  if (rc != APSS_OK) throw std::runtime_error(std::string("apss_create: ") + apss_last_error(nullptr));
}

GpuIndexingWorker::~GpuIndexingWorker() {
  if (g_) apss_group_destroy(g_);
  else apss_destroy(h_);
}

int64_t GpuIndexingWorker::storedVectors() const {
  int64_t rows = 0;
  if (g_) {
    apss_group_stats st{};
    st.struct_size = (int32_t)sizeof(st);
    if (apss_group_stats_get(g_, &st) == APSS_OK) rows = st.rows;
  } else {
    apss_size(h_, &rows, nullptr);
  }
  return rows;
}

SimilarityOutput GpuIndexingWorker::handle(const IndexData &m) {
  // flatten the batch to CSR; every vector's size must equal vectorDim (the require of CommonUtils.scala:99)
  std::vector<int64_t> rowptr{0}, ids;
  std::vector<int32_t> idx;
  std::vector<double> val;
  for (const auto &kv : m.vectors) {
    const SparseVector &v = kv.second;
    if (v.size != conf_.vectorDim)
      throw std::invalid_argument("requirement failed: vector1 size: " + std::to_string(v.size) + ", vector2 size: " +
                                  std::to_string(conf_.vectorDim));
    auto it = id_of_.find(kv.first);
    if (it == id_of_.end()) {
      it = id_of_.emplace(kv.first, (int64_t)name_of_.size()).first;
      name_of_.push_back(kv.first);
    }
    ids.push_back(it->second);
    idx.insert(idx.end(), v.indices.begin(), v.indices.end());
    val.insert(val.end(), v.values.begin(), v.values.end());
    rowptr.push_back((int64_t)idx.size());
  }
  int64_t n_res = 0;
  const int64_t n = (int64_t)ids.size();
  int32_t rc;
  if (g_)
    rc = stop_update_index_  // IndexingWorkerActor.scala:125-133
             ? apss_group_query(g_, n, rowptr.data(), idx.data(), val.data(), ids.data(), &n_res)
             : apss_group_insert_and_query(g_, n, rowptr.data(), idx.data(), val.data(), ids.data(), &n_res);
  else
    rc = stop_update_index_
             ? apss_query(h_, n, rowptr.data(), idx.data(), val.data(), ids.data(), &n_res)
             : apss_insert_and_query(h_, n, rowptr.data(), idx.data(), val.data(), ids.data(), &n_res);
  if (rc != APSS_OK) throw std::runtime_error(g_ ? apss_group_last_error(g_) : apss_last_error(h_));
  std::vector<int64_t> q((size_t)n_res), c((size_t)n_res);
  std::vector<float> s((size_t)n_res);
  if (n_res && (g_ ? apss_group_fetch_results(g_, 0, n_res, q.data(), c.data(), s.data())
                   : apss_fetch_results(h_, 0, n_res, q.data(), c.data(), s.data())) != APSS_OK)
    throw std::runtime_error(g_ ? apss_group_last_error(g_) : apss_last_error(h_));
  SimilarityOutput out;
  for (const auto &kv : m.vectors) out.output[kv.first];  // every query gets an entry, possibly empty (:106-107)
  for (int64_t i = 0; i < n_res; ++i) out.output[name_of_[(size_t)q[i]]][name_of_[(size_t)c[i]]] = (double)s[i];
  out.outputMoment = now_ms();
  return out;
}

void GpuIndexingWorker::receive(const IndexData &m) {
  try {
    SimilarityOutput out = handle(m);
    if (!reply_to_) return;
    if (conf_.outputIODuration <= 0) {
      reply_to_(out);  // replyTo.get ! SimilarityOutput(...), :130
    } else {           // updateWriteBuffer, :113-120
      for (auto &q : out.output)
        for (auto &c : q.second) write_buffer_[q.first][c.first] = c.second;
    }
  } catch (const std::exception &e) {  // case e: Exception => e.printStackTrace(), :135-137: the batch's output is lost
    last_error_ = e.what();
    std::fprintf(stderr, "GpuIndexingWorker: %s\n", e.what());
  }
}

void GpuIndexingWorker::receive(const IOTicket &) {
  if (write_buffer_.empty()) return;
  SimilarityOutput out;
  out.output = write_buffer_;  // writeBuffer.clone()
  out.outputMoment = now_ms();
  write_buffer_.clear();
  if (reply_to_) reply_to_(out);
}

void GpuIndexingWorker::receive(const Test &t) {
  std::printf("receiving Test(%s) in IndexWorkerActor\n", t.content.c_str());
}

void GpuIndexingWorker::receiveTimeout() { stop_update_index_ = true; }

void Region::tell(const VectorIOMsg &m) {
  IndexData d;
  d.vectors = m.vectors;
  worker_->receive(d);
}

ClientConnection::ClientConnection(const std::vector<std::string> &remoteAddresses,
                                   std::function<std::shared_ptr<Region>(const std::string &)> resolve) {
  for (const std::string &hp : remoteAddresses) {  // ClientConnection.scala:12-21
    const size_t colon = hp.find(':');
    if (colon == std::string::npos) throw std::out_of_range("ArrayIndexOutOfBoundsException: 1");  // hostAndPort(1)
    routers_.push_back("akka.tcp://ClusterSystem@" + hp.substr(0, colon) + ":" + hp.substr(colon + 1) + "/user/regionRouter");
  }
  if (routers_.empty()) throw std::invalid_argument("bound must be positive");  // Random.nextInt(0)
  std::mt19937 rng{std::random_device{}()};
  remote_router_ = resolve(routers_[rng() % routers_.size()]);  // Random.nextInt(remoteAddresses.length), :23-24
}

void ClientConnection::insertNewVector(const std::vector<std::pair<std::string, SparkSparseVector>> &vectors) {
  VectorIOMsg m;
  m.vectors = vectors;
  if (remote_router_) remote_router_->tell(m);  // remoteRouter ! VectorIOMsg(vectors)
}

}  // namespace cpslab
