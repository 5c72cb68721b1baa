// tests/mock/exchange_mock.cpp -- TEST INFRASTRUCTURE: the product's exchange posting order (csrc/sg_exchange_order.hpp, the very template
// csrc/sg_rowband_rccl.hip instantiates with RCCL) on a transport made of callbacks, so that tests/test_exchange_order_gloo.py can run it
// between CPU ranks over gloo.  Built by the test with g++; nothing in the product links it.
#include <cstddef>

#include "sg_exchange_order.hpp"

extern "C" {
typedef int (*sg_mock_group_fn)(void);
typedef int (*sg_mock_xfer_fn)(void *buf, size_t words, int peer);

struct SgMockTransport {
    sg_mock_group_fn start, end;
    sg_mock_xfer_fn send_fn, recv_fn;
    bool group_start() { return start() == 0; }
    bool group_end() { return end() == 0; }
    bool send(const void *p, size_t words, int peer) { return send_fn(const_cast<void *>(p), words, peer) == 0; }
    bool recv(void *p, size_t words, int peer) { return recv_fn(p, words, peer) == 0; }
};

int sg_mock_exchange(int peer_a, int peer_b, const void *send_a, const void *send_b, void *recv_a, void *recv_b, size_t words,
                     sg_mock_group_fn start, sg_mock_group_fn end, sg_mock_xfer_fn send_fn, sg_mock_xfer_fn recv_fn)
{
    SgMockTransport t{start, end, send_fn, recv_fn};
    return sg::exchange_post(t, peer_a, peer_b, send_a, send_b, recv_a, recv_b, words);
}
}
