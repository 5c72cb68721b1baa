// sg_exchange_order.hpp -- the POSTING ORDER of one halo exchange, separated from its transport.
//
// csrc/sg_rowband_rccl.hip instantiates it with RCCL (ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd); tests/mock/exchange_mock.cpp
// instantiates the same template with callbacks, and tests/test_exchange_order_gloo.py drives that build with gloo point-to-point between
// two and three CPU ranks -- so the two-distinct-neighbours branch, which RCCL itself has only ever run on a one-rank communicator here
// (these boxes have one GPU), is at least executed with real messages.
//
// A rank sends its FIRST block towards peer_a and its LAST block towards peer_b, and receives into recv_a what a sends towards it (a's last
// block) and into recv_b b's first block.  Messages between one pair of ranks match in posting order.  When both peers are the same rank
// (a ring of two, or one rank talking to itself) that rank posts send(first), send(last) too, and what arrives first is ITS first block --
// which is this rank's recv_b.  So the receives are posted b first in that case, a first otherwise.
#pragma once

#include <cstddef>

namespace sg {

template <class Transport>
int exchange_post(Transport &t, int peer_a, int peer_b, const void *send_a, const void *send_b, void *recv_a, void *recv_b, size_t words)
{
    const bool a = peer_a >= 0, b = peer_b >= 0;
    if (!a && !b) return 0;
    if (!t.group_start()) return -1;
    bool ok = true;
    if (a) ok = ok && t.send(send_a, words, peer_a);
    if (b) ok = ok && t.send(send_b, words, peer_b);
    if (a && b && peer_a == peer_b) {
        ok = ok && t.recv(recv_b, words, peer_b) && t.recv(recv_a, words, peer_a);
    } else {
        if (a) ok = ok && t.recv(recv_a, words, peer_a);
        if (b) ok = ok && t.recv(recv_b, words, peer_b);
    }
    if (!t.group_end()) return -1;
    return ok ? 0 : -1;
}

}  // namespace sg
