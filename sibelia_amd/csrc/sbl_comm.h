// sbl_comm.h -- the communicator a context carries when several GPUs work on one job (shard.hip implements the transports:
// RCCL -- grouped ncclSend / ncclRecv + ncclAllGather on the context's stream -- and a local one for contexts of one process).
// Two primitives are all the sharded pipelines need (hash-prefix sharded k-mer table: shard.hip; sharded rank doubling: longk.hip).
#pragma once
#include <stddef.h>
#include <stdint.h>

struct sbl_ctx;

struct SblComm {
	uint32_t rank = 0, n = 1;
	virtual ~SblComm() {}
	// small host payloads (counts): out = n x bytes
	virtual void allgather_host(sbl_ctx *c, const void *in, size_t bytes, void *out) = 0;
	// device buffers; byte counts / offsets per peer
	virtual void alltoallv(sbl_ctx *c, const char *send, const size_t *sbytes, const size_t *soff,
	                       char *recv, const size_t *rbytes, const size_t *roff) = 0;
	// this rank is leaving a collective call with an error: release peers that would wait for it (local transport)
	virtual void abort_peers() {}
};
