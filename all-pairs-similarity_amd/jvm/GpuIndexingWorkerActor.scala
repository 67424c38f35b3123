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
  * (maxIndexEntryActorNum = 1, maxShardNum = 1), since every worker of the reference recomputes the same full score anyway.
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
  // which GPU this worker's index lives on (one handle = one GPU's HBM); GPU keys of this build: cpslab.allpair.gpu.*
  private val device = if (conf.hasPath("cpslab.allpair.gpu.device")) conf.getInt("cpslab.allpair.gpu.device") else 0
  private val headTerms = if (conf.hasPath("cpslab.allpair.gpu.headTerms")) conf.getInt("cpslab.allpair.gpu.headTerms") else 0
  private val handle = NativeApss.create(vectorDim, similarityThreshold, indexThreshold,
    if (indexThreshold > 0.0) NativeApss.FLAG_VALUE_PRUNE else 0, device, headTerms)
  require(handle != 0L, NativeApss.lastError(0L))
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

  override def postStop(): Unit = NativeApss.destroy(handle)

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
    val n = NativeApss.submit(handle, if (stopUpdateIndex) 1 else 2, rowptr, indices, values, ids)
    if (n < 0) throw new IllegalArgumentException(NativeApss.lastError(handle))
    val q = new Array[Long](n.toInt); val c = new Array[Long](n.toInt); val s = new Array[Float](n.toInt)
    if (n > 0) NativeApss.fetch(handle, n, q, c, s)
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
