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
  /** name the dense-head block's terms instead of the library's policy (include/apss.h, apss_set_head_terms; a handle made by
    * `create` holds the whole term space: part = 0, nParts = 1); empty handle only.  Returns 0 or a negative status.
    * (Term-sharded deployments do not call this: a group -- createGroup -- gives its members their shares itself.) */
  @native def setHeadTerms(h: Long, terms: Array[Int], part: Int, nParts: Int): Int
  /** how many of the 256 columns of a head of more than 256 terms are FOLDED columns: 64 | 128 | 192 (0 = the default, 128);
    * the other 256 - columns most frequent terms keep a column each.  Takes effect at the next setHeadTerms */
  @native def setHeadFold(h: Long, columns: Int): Int
  /** the block's terms, chosen by the library or set by setHeadTerms (what one shard's policy decided is what its peers are given) */
  @native def headTerms(h: Long): Array[Int]

  // ---- apss_group: the term-sharded index of one node behind one object (include/apss.h).  Member i lives on devices(i) and
  // owns a contiguous term range cut on the first batch; every batch is handed WHOLE to every member; the exchange of the
  // members' answers (all-gather of candidate lists, RCCL all-reduce of per-candidate partial scores) runs below this call.
  val GROUP_FORCE_EXCHANGE = 1
  val GROUP_NO_RCCL = 2
  /** returns the group's handle or 0 (groupLastError(0) says why) */
  @native def createGroup(dim: Int, theta: Double, indexThreshold: Double, flags: Int, devices: Array[Int], headTerms: Int,
                          groupFlags: Int): Long
  @native def destroyGroup(g: Long): Unit
  @native def groupLastError(g: Long): String
  /** mode 0 insert, 1 query on the frozen index, 2 insert-and-query; returns #results or a negative status */
  @native def groupSubmit(g: Long, mode: Int, rowptr: Array[Long], indices: Array[Int], values: Array[Double],
                          ids: Array[Long]): Long
  @native def groupFetch(g: Long, count: Long, outQ: Array[Long], outC: Array[Long], outScore: Array[Float]): Int
  /** out(0..8) = members, exchange (0 none, 1 copies, 2 RCCL), head terms, rows, candidates summed over the members, distinct
    * candidates, result pairs, bytes all-gathered per member, bytes all-reduced -- of the last call */
  @native def groupStats(g: Long, out: Array[Long]): Int
}
