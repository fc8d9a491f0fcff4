// oracle/shim -- TEST INFRASTRUCTURE ONLY: placeholder (vk::RingBuffer is used off-path only)
#pragma once
