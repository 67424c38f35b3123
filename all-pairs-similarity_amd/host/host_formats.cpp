// host_formats.cpp -- CPU-only driver of the data-format side of the host mirror (no GPU call is made):
//   host_formats ccweb < lines      one "id<TAB>(size,[i,...],[v,...])" per line, "ERROR" where the reference throws
//   host_formats vector < lines     SparseVector.fromString / toString round trip
//   host_formats ccweb-file PATH    CCWEBVideoLoadGenerator(path).generateVectors
//   host_formats tfidf NUM_FEATURES FILE...   the ETL's HashingTF + IDF (+ L2 normalisation), one vector per file
#include <cstdio>
#include <iostream>
#include <string>
#include <vector>

#include "cpslab_host.hpp"

using namespace cpslab;

int main(int argc, char **argv) {
  const std::string mode = argc > 1 ? argv[1] : "";
  if (mode == "ccweb-file" && argc > 2) {
    try {
      for (const auto &kv : CCWEBVideoLoadGenerator(argv[2]).generateVectors())
        std::printf("%s\t%s\n", kv.first.c_str(), kv.second.toString().c_str());
    } catch (const std::exception &) {
      std::printf("ERROR\n");
    }
    return 0;
  }
  if (mode == "tfidf" && argc > 2) {  // tfidf NUM_FEATURES FILE...  (files in the order given = document order)
    try {
      const int nf = std::stoi(argv[2]);
      std::vector<SparseVector> tf;
      for (int i = 3; i < argc; ++i) tf.push_back(etl::hashingTF(etl::documentTokens(argv[i]), nf));
      for (const auto &v : etl::tfidf(tf, true)) std::printf("%s\n", v.toString().c_str());
    } catch (const std::exception &e) {
      std::printf("ERROR %s\n", e.what());
    }
    return 0;
  }
  if (mode != "ccweb" && mode != "vector") {
    std::fprintf(stderr, "usage: host_formats ccweb|vector < lines | host_formats ccweb-file PATH\n");
    return 2;
  }
  std::string line;
  while (std::getline(std::cin, line)) {
    try {
      if (mode == "ccweb") {
        const auto kv = CCWEBVideoLoadGenerator::lineParser(line);
        std::printf("%s\t%s\n", kv.first.c_str(), kv.second.toString().c_str());
      } else {
        std::printf("%s\n", SparseVector::fromString(line).toString().c_str());
      }
    } catch (const std::exception &) {
      std::printf("ERROR\n");
    }
  }
  return 0;
}
