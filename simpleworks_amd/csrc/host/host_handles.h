// Host-only handles of the C ABI (include/swmarlin.h): a generator and a verifying key.
#pragma once
#include "ahp.h"

struct swm_rng {
    swm::ChaChaRng r;
};
struct swm_vk {
    swm::VerifyingKey vk;
};
