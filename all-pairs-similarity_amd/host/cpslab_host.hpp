// cpslab_host.hpp -- C++ mirror of the reference's host-side interface for the hot path, above the C ABI.
//
// The reference is Scala on the JVM and no JDK exists in this image, so the classes a user of the reference touches
// on this path are mirrored here in C++ with the same names, argument meaning and error behaviour (the Scala/JNI
// originals a maintainer would actually deploy are under ../jvm/):
//   cpslab::SparseVector           core/src/main/scala/cpslab/vector/SparseVector.scala:198-223 (+ parser :132-141)
//   cpslab::VectorIOMsg/IndexData/SimilarityOutput/Test/IOTicket   .../message/Message.scala:13-43
//   cpslab::GpuIndexingWorker      .../deploy/server/IndexingWorkerActor.scala:21-148 with the index on the GPU
//   cpslab::ClientConnection       .../deploy/client/ClientConnection.scala:10-33
// Akka itself (remoting, sharding, routers) is out of scope: "actors" are plain objects, `!` is a direct call.
#pragma once
#include <cstdint>
#include <functional>
#include <memory>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

struct apss_handle;
struct apss_group;

namespace cpslab {

// SparseVector(size, indices, values); `require(indices.length == values.length)` (SparseVector.scala:202)
struct SparseVector {
  int size = 0;
  std::vector<int32_t> indices;
  std::vector<double> values;
  SparseVector() = default;
  SparseVector(int size_, std::vector<int32_t> idx, std::vector<double> val);
  std::string toString() const;                          // "(size,[i,...],[v,...])", SparseVector.scala:204-205
  static SparseVector fromString(const std::string &s);  // Vectors.fromString, SparseVector.scala:132-141
};
using SparkSparseVector = SparseVector;  // org.apache.spark.mllib.linalg.SparseVector is used as the same struct

// benchmark/CCWEBVideoLoadGenerator.scala:8-30 -- the load generator of the CC_WEB_VIDEO feature dump: one
// "(id,(size,[...],[v0,...,v{size-1}]))"-shaped line per video; the LAST `size` comma-separated fields are the dense
// feature values, zeros are dropped (CCW:15-20).  Feeds ClientConnection.insertNewVector.
class CCWEBVideoLoadGenerator {
 public:
  explicit CCWEBVideoLoadGenerator(std::string path) : path_(std::move(path)) {}
  // private def lineParser(line: String): (String, SparkSparseVector), CCW:10-21; throws std::invalid_argument where the
  // reference throws NumberFormatException / IndexOutOfBounds
  static std::pair<std::string, SparkSparseVector> lineParser(const std::string &line);
  // def generateVectors: List[(String, SparkSparseVector)], CCW:23-29
  std::vector<std::pair<std::string, SparkSparseVector>> generateVectors() const;

 private:
  std::string path_;
};

// ---- TF-IDF ingest of a text corpus (BASELINE config 1): etl/src/main/scala/cpslab/etl/PreprocessWithTFIDF.scala:21-52 with
// Spark 1.2.0 mllib's HashingTF / IDF formulas (formula-level restatement: the Spark sources are not in the reference tree)
namespace etl {
int32_t javaStringHashCode(const std::string &s);  // java.lang.String.hashCode over ISO-8859-1 chars (one per byte)
int32_t nonNegativeMod(int32_t x, int32_t mod);    // org.apache.spark.util.Utils.nonNegativeMod
// PreprocessWithTFIDF.scala:33-41: every line + " ", then the literal "null " of the read loop, split on " "
std::vector<std::string> documentTokens(const std::string &path);
// mllib.feature.HashingTF(numFeatures).transform: index = nonNegativeMod(term.hashCode, numFeatures), value = count
SparseVector hashingTF(const std::vector<std::string> &tokens, int32_t numFeatures = 1 << 20);
// mllib.feature.IDF().fit(tf).transform(tf): idf_t = ln((m + 1) / (df_t + 1)); optional L2 normalisation
// (benchmark/LoadGenerator.scala:34-37: the reference's client normalises, its ETL does not)
std::vector<SparseVector> tfidf(const std::vector<SparseVector> &tf, bool normalize);
}  // namespace etl

struct VectorIOMsg { std::vector<std::pair<std::string, SparkSparseVector>> vectors; };  // Message.scala:13
struct IndexData { std::vector<std::pair<std::string, SparseVector>> vectors; };         // Message.scala:18 (wrappers' payload)
struct Test { std::string content; };                                                    // Message.scala:37
struct IOTicket {};                                                                      // Message.scala:39
struct SimilarityOutput {                                                                // Message.scala:20-35
  std::unordered_map<std::string, std::unordered_map<std::string, double>> output;
  int64_t outputMoment = 0;  // System.currentTimeMillis
  std::string toString() const;
};

// the cpslab.allpair.* keys the path reads (IndexingWorkerActor.scala:23-33, WriteWorkerActor.scala:35, EntryProxyActor.scala:25)
struct Config {
  double similarityThreshold = 0.0;
  int vectorDim = 0;
  int64_t outputIODuration = 0;   // <= 0: reply per batch; > 0: accumulate, flush on IOTicket
  double indexThreshold = 0.0;
  bool applyIndexThreshold = false;
  int deviceId = 0;
  int tileRows = 0;
  // cpslab.allpair.gpu.devices (GpuIndexingWorkerActor.scala): more than one entry = the term-sharded index of the node, one
  // member per listed GPU (apss_group, include/apss.h); empty or one entry = one handle on deviceId / that device
  std::vector<int> devices;
  int headTerms = 0;          // cpslab.allpair.gpu.headTerms
  unsigned groupFlags = 0;    // APSS_GROUP_* (tests: the RCCL exchange with one member)
};

// IndexingWorkerActor with vectorsStore / invertedIndex resident on the GPU.
class GpuIndexingWorker {
 public:
  using ReplyTo = std::function<void(const SimilarityOutput &)>;  // the actor at cpslab.allpair.outputActor
  GpuIndexingWorker(const Config &conf, ReplyTo replyTo);
  ~GpuIndexingWorker();
  GpuIndexingWorker(const GpuIndexingWorker &) = delete;
  GpuIndexingWorker &operator=(const GpuIndexingWorker &) = delete;

  void receive(const IndexData &m);  // IndexingWorkerActor.scala:123-137: exceptions are printed and swallowed
  void receive(const IOTicket &);    // :138-142
  void receive(const Test &t);       // :145-147 (echo)
  void receiveTimeout();             // ReceiveTimeout -> stopUpdateIndex = true, :143-144
  int64_t storedVectors() const;
  const std::string &lastError() const { return last_error_; }

 private:
  SimilarityOutput handle(const IndexData &m);
  Config conf_;
  ReplyTo reply_to_;
  apss_handle *h_ = nullptr;
  apss_group *g_ = nullptr;   // set instead of h_ when conf.devices names several GPUs (or groupFlags force the exchange)
  bool stop_update_index_ = false;
  std::unordered_map<std::string, int64_t> id_of_;  // String id <-> the ABI's int64 handle
  std::vector<std::string> name_of_;
  std::unordered_map<std::string, std::unordered_map<std::string, double>> write_buffer_;  // :27, 113-120
  std::string last_error_;
};

// The region the client's router resolves to.  In the reference this is Akka remoting + cluster sharding +
// EntryProxyActor/WriteWorkerActor fan-out (out of scope); here it hands the batch to one GPU worker.
class Region {
 public:
  explicit Region(std::shared_ptr<GpuIndexingWorker> w) : worker_(std::move(w)) {}
  void tell(const VectorIOMsg &m);
 private:
  std::shared_ptr<GpuIndexingWorker> worker_;
};

// class ClientConnection(remoteAddresses: List[String], localActorSystem) -- ClientConnection.scala:10
class ClientConnection {
 public:
  // "host:port" strings are parsed exactly like the reference (split(":"), two fields) and turned into
  // akka.tcp://ClusterSystem@host:port/user/regionRouter paths; `resolve` maps such a path to a Region.
  ClientConnection(const std::vector<std::string> &remoteAddresses,
                   std::function<std::shared_ptr<Region>(const std::string &routerPath)> resolve);
  // def insertNewVector(vectors: Set[(String, SparkSparseVector)]): Unit -- fire and forget (ClientConnection.scala:31-33)
  void insertNewVector(const std::vector<std::pair<std::string, SparkSparseVector>> &vectors);
  const std::vector<std::string> &remoteAddressRouterList() const { return routers_; }
 private:
  std::vector<std::string> routers_;
  std::shared_ptr<Region> remote_router_;
};

}  // namespace cpslab
