// host_selftest.cpp -- the SURVEY section-3.3 known-answer test driven through the mirrored reference interface:
// ClientConnection.insertNewVector -> (region) -> GpuIndexingWorker.receive(IndexData) -> SimilarityOutput.
// Needs a GPU.  Build: see Makefile in this directory.
#include <cmath>
#include <cstdio>
#include <map>

#include "cpslab_host.hpp"

using namespace cpslab;

static int fails = 0;
#define CHECK(c)                                              \
  do {                                                        \
    if (!(c)) {                                               \
      std::printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #c); \
      ++fails;                                                \
    }                                                         \
  } while (0)

int main() {
  // text format round trip (SparseVector.scala:132-141, 204-205)
  SparseVector p = SparseVector::fromString("(10,[1,4,7],[0.5,0.25,2.0])");
  CHECK(p.size == 10 && p.indices.size() == 3 && p.values[2] == 2.0);
  CHECK(SparseVector::fromString(p.toString()).values == p.values);

  Config conf;
  conf.similarityThreshold = 0.5;
  conf.vectorDim = 4;
  conf.tileRows = 64;
  std::vector<SimilarityOutput> got;
  auto worker = std::make_shared<GpuIndexingWorker>(conf, [&](const SimilarityOutput &o) { got.push_back(o); });
  auto region = std::make_shared<Region>(worker);
  ClientConnection conn({"127.0.0.1:2551"}, [&](const std::string &path) {
    CHECK(path == "akka.tcp://ClusterSystem@127.0.0.1:2551/user/regionRouter");
    return region;
  });

  // streaming KAT, intended semantics: batch1 = {v1, v3}, batch2 = {v2, v5}
  conn.insertNewVector({{"v1", SparseVector(4, {0, 2}, {0.6, 0.8})}, {"v3", SparseVector(4, {2}, {1.0})}});
  conn.insertNewVector({{"v2", SparseVector(4, {0, 2}, {0.6, 0.8})}, {"v5", SparseVector(4, {0}, {1.0})}});
  CHECK(got.size() == 2);
  auto near = [](double a, double b) { return std::fabs(a - b) < 1e-6; };
  if (got.size() == 2) {
    auto &b1 = got[0].output, &b2 = got[1].output;
    CHECK(b1.size() == 2 && b1["v1"].size() == 1 && near(b1["v1"]["v3"], 0.8) && near(b1["v3"]["v1"], 0.8));
    CHECK(b2["v2"].size() == 3 && near(b2["v2"]["v1"], 1.0) && near(b2["v2"]["v3"], 0.8) && near(b2["v2"]["v5"], 0.6));
    CHECK(b2["v5"].size() == 2 && near(b2["v5"]["v1"], 0.6) && near(b2["v5"]["v2"], 0.6));
    CHECK(got[1].outputMoment >= got[0].outputMoment && got[0].outputMoment > 0);
  }
  CHECK(worker->storedVectors() == 4);

  // a vector of the wrong size: the handler swallows the exception, the batch's output is lost (IndexingWorkerActor.scala:135-137)
  conn.insertNewVector({{"bad", SparseVector(5, {0}, {1.0})}});
  CHECK(got.size() == 2 && worker->storedVectors() == 4 && !worker->lastError().empty());

  // frozen index after ReceiveTimeout: query only
  worker->receiveTimeout();
  conn.insertNewVector({{"v6", SparseVector(4, {2, 3}, {0.8, 0.6})}});
  CHECK(got.size() == 3 && worker->storedVectors() == 4);
  if (got.size() == 3) CHECK(got[2].output["v6"].size() == 3 && near(got[2].output["v6"]["v3"], 0.8));

  // buffered output (outputIODuration > 0): nothing until the IOTicket
  Config c2 = conf;
  c2.outputIODuration = 50;
  std::vector<SimilarityOutput> got2;
  GpuIndexingWorker w2(c2, [&](const SimilarityOutput &o) { got2.push_back(o); });
  IndexData d;
  d.vectors = {{"a", SparseVector(4, {1}, {1.0})}, {"b", SparseVector(4, {1}, {1.0})}};
  w2.receive(d);
  CHECK(got2.empty());
  w2.receive(IOTicket{});
  CHECK(got2.size() == 1 && near(got2[0].output["a"]["b"], 1.0));
  w2.receive(IOTicket{});
  CHECK(got2.size() == 1);  // empty buffer: no message

  // the same streaming KAT through the term-sharded index of the node (cpslab.allpair.gpu.devices): two members sharing GPU 0
  // (exchange by copies), then one member running the RCCL exchange itself (APSS_GROUP_FORCE_EXCHANGE = 1)
  for (int variant = 0; variant < 2; ++variant) {
    Config cg = conf;
    if (variant == 0) cg.devices = {0, 0};
    else { cg.devices = {0}; cg.groupFlags = 1u; }
    std::vector<SimilarityOutput> gg;
    GpuIndexingWorker wg(cg, [&](const SimilarityOutput &o) { gg.push_back(o); });
    IndexData b1, b2, b3;
    b1.vectors = {{"v1", SparseVector(4, {0, 2}, {0.6, 0.8})}, {"v3", SparseVector(4, {2}, {1.0})}};
    b2.vectors = {{"v2", SparseVector(4, {0, 2}, {0.6, 0.8})}, {"v5", SparseVector(4, {0}, {1.0})}};
    wg.receive(b1);
    wg.receive(b2);
    CHECK(gg.size() == 2 && wg.lastError().empty());
    if (gg.size() == 2) {
      auto &o1 = gg[0].output, &o2 = gg[1].output;
      CHECK(o1.size() == 2 && o1["v1"].size() == 1 && near(o1["v1"]["v3"], 0.8) && near(o1["v3"]["v1"], 0.8));
      CHECK(o2["v2"].size() == 3 && near(o2["v2"]["v1"], 1.0) && near(o2["v2"]["v3"], 0.8) && near(o2["v2"]["v5"], 0.6));
      CHECK(o2["v5"].size() == 2 && near(o2["v5"]["v1"], 0.6) && near(o2["v5"]["v2"], 0.6));
    }
    CHECK(wg.storedVectors() == 4);
    wg.receiveTimeout();
    b3.vectors = {{"v6", SparseVector(4, {2, 3}, {0.8, 0.6})}};
    wg.receive(b3);
    CHECK(gg.size() == 3 && wg.storedVectors() == 4);
    if (gg.size() == 3) CHECK(gg[2].output["v6"].size() == 3 && near(gg[2].output["v6"]["v3"], 0.8));
  }

  std::printf(fails ? "host_selftest: %d FAILED\n" : "host_selftest: PASS\n", fails);
  return fails ? 1 : 0;
}
