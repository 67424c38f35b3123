package cpslab.gpu

/** JNI face of libapss_hip.so (include/apss.h); see all-pairs-similarity_amd/jvm/apss_jni.c.
  * Not compiled in the build image (no JVM there); it is the binding a maintainer adds to `core/`. */
object NativeApss {
  System.loadLibrary("apss_jni")
  val FLAG_VALUE_PRUNE = 1
  val FLAG_ADMISSION = 2
  val FLAG_NORMALIZE = 4
  val FLAG_NO_SYMMETRY = 64 // a batch that is the whole store is probed in both directions, as IndexingWorkerActor does (include/apss.h)
  /** headTerms: dense-head block of the library (0 = it decides from the term distribution, -1 = never) */
  @native def create(dim: Int, theta: Double, indexThreshold: Double, flags: Int, device: Int, headTerms: Int): Long
  @native def destroy(h: Long): Unit
  @native def lastError(h: Long): String
  /** mode 0 insert, 1 query on the frozen index, 2 insert-and-query; returns #results or a negative status */
  @native def submit(h: Long, mode: Int, rowptr: Array[Long], indices: Array[Int], values: Array[Double],
                     ids: Array[Long]): Long
  @native def fetch(h: Long, count: Long, outQ: Array[Long], outC: Array[Long], outScore: Array[Float]): Int
  /** term-sharded deployments (one worker per GPU owning a term range): the join's dense-head block, the same terms on
    * every shard, worker `part` of `nParts` multiplying its share of the candidate tiles (include/apss.h,
    * apss_set_head_terms); empty handle only.  Returns 0 or a negative status */
  @native def setHeadTerms(h: Long, terms: Array[Int], part: Int, nParts: Int): Int
  /** columns (128 | 256, 0 = default) of the folded block of a head of more than 256 terms: what the handle that chose the
    * terms justified on its sample (headColumns - 256); before setHeadTerms */
  @native def setHeadFold(h: Long, columns: Int): Int
  /** the block's terms, chosen by the library or set by setHeadTerms (what one shard's policy decided is what its peers are given) */
  @native def headTerms(h: Long): Array[Int]
}
