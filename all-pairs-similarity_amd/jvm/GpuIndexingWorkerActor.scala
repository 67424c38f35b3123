package cpslab.deploy.server

import scala.collection.mutable
import scala.concurrent.duration._
import scala.language.postfixOps

import akka.actor.{Actor, ActorSelection, Cancellable, ReceiveTimeout}
import com.typesafe.config.Config
import cpslab.gpu.NativeApss
import cpslab.message._

/** Drop-in for IndexingWorkerActor (core/src/main/scala/cpslab/deploy/server/IndexingWorkerActor.scala): same
  * messages in (IndexData, IOTicket, ReceiveTimeout, Test), same SimilarityOutput out, same config keys; the
  * vectorsStore / invertedIndex / calculateSimilarity loop is replaced by one JNI call per batch.
  * EntryProxyActor.handleDataPacket keeps sending it IndexData; with the GPU index a single worker per entry is enough
  * (maxIndexEntryActorNum = 1, maxShardNum = 1), since every worker of the reference recomputes the same full score anyway;
  * the term sharding of a multi-GPU node happens INSIDE the library (cpslab.allpair.gpu.devices, below).
  * Source only: not compiled in the build image (no JVM there). */
private class GpuIndexingWorkerActor(conf: Config) extends Actor {
  val similarityThreshold = conf.getDouble("cpslab.allpair.similarityThreshold")
  val vectorDim = conf.getInt("cpslab.allpair.vectorDim")
  val outputWritingDuration = conf.getLong("cpslab.allpair.outputIODuration")
  val writeBuffer = new mutable.HashMap[String, mutable.HashMap[String, Double]]
  private val expDuration = conf.getLong("cpslab.allpair.benchmark.expDuration")
  // WriteWorkerActor.scala:35,188-194 drops entries <= indexThreshold before a vector reaches this actor; the key is read here
  // too so that a deployment that feeds the actor directly gets the same pruning on the device
  private val indexThreshold =
    if (conf.hasPath("cpslab.allpair.indexThreshold")) conf.getDouble("cpslab.allpair.indexThreshold") else 0.0
  // Where the index lives.  cpslab.allpair.gpu.devices = [0, 1, .., 7]: the term-sharded index of the node -- one member per
  // listed GPU owning a contiguous term range, the exchange of partial scores over RCCL below the JNI boundary (apss_group,
  // include/apss.h): the reference's own fan-out of a DataPacket to maxShardNum x maxIndexEntryActorNum term workers
  // (WriteWorkerActor.scala:164-183, EntryProxyActor.scala:37-49) done inside the library.  One entry, or only
  // cpslab.allpair.gpu.device: one handle in one GPU's HBM.
  private val devices: Array[Int] =
    if (conf.hasPath("cpslab.allpair.gpu.devices")) {
      val l = conf.getIntList("cpslab.allpair.gpu.devices"); Array.tabulate(l.size)(i => l.get(i).intValue)
    } else Array(if (conf.hasPath("cpslab.allpair.gpu.device")) conf.getInt("cpslab.allpair.gpu.device") else 0)
  private val headTerms = if (conf.hasPath("cpslab.allpair.gpu.headTerms")) conf.getInt("cpslab.allpair.gpu.headTerms") else 0
  private val flags = if (indexThreshold > 0.0) NativeApss.FLAG_VALUE_PRUNE else 0
  private val grouped = devices.length > 1
  private val handle =
    if (grouped) NativeApss.createGroup(vectorDim, similarityThreshold, indexThreshold, flags, devices, headTerms, 0)
    else NativeApss.create(vectorDim, similarityThreshold, indexThreshold, flags, devices(0), headTerms)
  require(handle != 0L, if (grouped) NativeApss.groupLastError(0L) else NativeApss.lastError(0L))
  private def submit(mode: Int, rowptr: Array[Long], indices: Array[Int], values: Array[Double], ids: Array[Long]): Long =
    if (grouped) NativeApss.groupSubmit(handle, mode, rowptr, indices, values, ids)
    else NativeApss.submit(handle, mode, rowptr, indices, values, ids)
  private def fetch(n: Long, q: Array[Long], c: Array[Long], s: Array[Float]): Int =
    if (grouped) NativeApss.groupFetch(handle, n, q, c, s) else NativeApss.fetch(handle, n, q, c, s)
  private def lastError: String = if (grouped) NativeApss.groupLastError(handle) else NativeApss.lastError(handle)
  private val idOf = new mutable.HashMap[String, Long]
  private val nameOf = new mutable.ArrayBuffer[String]
  private var stopUpdateIndex = false
  var replyTo: Option[ActorSelection] = None
  var ioTask: Cancellable = null

  if (expDuration > 0) context.setReceiveTimeout(expDuration milliseconds)

  override def preStart(): Unit = {
    import context.dispatcher
    replyTo = Some(context.actorSelection(conf.getString("cpslab.allpair.outputActor")))
    if (outputWritingDuration > 0) {
      ioTask = context.system.scheduler.schedule(0 milliseconds, outputWritingDuration milliseconds, self, IOTicket)
    }
  }

  override def postStop(): Unit = if (grouped) NativeApss.destroyGroup(handle) else NativeApss.destroy(handle)

  private def runBatch(vectors: Set[cpslab.vector.SparseVectorWrapper]):
      mutable.HashMap[String, mutable.HashMap[String, Double]] = {
    // The whole vector is indexed and scored here, so it must arrive ONCE: with maxShardNum > 1 the reference sends the
    // same full vector in one DataPacket per shard, each naming that shard's dims in wrapper.indices
    // (WriteWorkerActor.scala:166-179), and it would be stored and queried once per packet.  Deploy with
    // cpslab.allpair.maxShardNum = 1 (and maxIndexEntryActorNum = 1): INTEGRATION.md.
    vectors.foreach(w => require(w.indices.size == w.sparseVector._2.indices.length,
      "GpuIndexingWorkerActor needs the whole vector in one DataPacket: set cpslab.allpair.maxShardNum = 1"))
    val batch = vectors.toArray.map(_.sparseVector)           // (String id, SparseVector)
    val rowptr = new Array[Long](batch.length + 1)
    val ids = new Array[Long](batch.length)
    for (i <- batch.indices) {
      require(batch(i)._2.size == vectorDim, s"vector1 size: ${batch(i)._2.size}, vector2 size: $vectorDim")
      rowptr(i + 1) = rowptr(i) + batch(i)._2.indices.length
      ids(i) = idOf.getOrElseUpdate(batch(i)._1, { nameOf += batch(i)._1; (nameOf.length - 1).toLong })
    }
    val indices = batch.flatMap(_._2.indices)
    val values = batch.flatMap(_._2.values)
    val n = submit(if (stopUpdateIndex) 1 else 2, rowptr, indices, values, ids)
    if (n < 0) throw new IllegalArgumentException(lastError)
    val q = new Array[Long](n.toInt); val c = new Array[Long](n.toInt); val s = new Array[Float](n.toInt)
    if (n > 0) fetch(n, q, c, s)
    val out = new mutable.HashMap[String, mutable.HashMap[String, Double]]
    batch.foreach(b => out.getOrElseUpdate(b._1, new mutable.HashMap[String, Double]))
    for (i <- 0 until n.toInt) out(nameOf(q(i).toInt)) += nameOf(c(i).toInt) -> s(i).toDouble
    out
  }

  def receive: Receive = {
    case IndexData(vectors) =>
      try {
        val out = runBatch(vectors)
        if (replyTo.isDefined) {
          if (outputWritingDuration <= 0) replyTo.get ! SimilarityOutput(out, System.currentTimeMillis())
          else for ((q, m) <- out; (c, s) <- m) writeBuffer.getOrElseUpdate(q, new mutable.HashMap[String, Double]) += c -> s
        }
      } catch {
        case e: Exception => e.printStackTrace()
      }
    case IOTicket =>
      if (writeBuffer.nonEmpty) {
        replyTo.get ! SimilarityOutput(writeBuffer.clone(), System.currentTimeMillis())
        writeBuffer.clear()
      }
    case ReceiveTimeout => stopUpdateIndex = true
    case t @ Test(_) => replyTo.get ! t
  }
}
