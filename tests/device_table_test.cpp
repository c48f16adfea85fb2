// CPU-side unit test of skeres_amd/csrc/device_table.hpp: the per-device table that owns the
// factorisation's queues.  Built and run by tests/test_capi_cpu.py with g++ (no HIP needed).
#include <atomic>
#include <cstdio>
#include <thread>
#include <vector>

#include "../skeres_amd/csrc/device_table.hpp"

struct Entry { int device; int serial; };

#define CHECK(c) do { if (!(c)) { std::fprintf(stderr, "device_table_test: %s failed (line %d)\n", #c, __LINE__); return 1; } } while (0)

int main() {
  sk::PerDeviceTable<Entry> table;
  std::atomic<int> created{0};
  auto make = [&](int dev) { return new Entry{dev, created.fetch_add(1)}; };
  // one entry per device, stable addresses, created once
  Entry* a = table.get_or_create(0, make);
  Entry* b = table.get_or_create(1, make);
  CHECK(a && b && a != b && a->device == 0 && b->device == 1);
  CHECK(table.get_or_create(0, make) == a && table.get_or_create(1, make) == b && created == 2 && table.size() == 2);
  CHECK(table.find(0) == a && table.find(7) == nullptr);
  // a failed creation stores nothing and is retried
  int attempts = 0;
  auto flaky = [&](int dev) -> Entry* { return ++attempts < 2 ? nullptr : new Entry{dev, -1}; };
  CHECK(table.get_or_create(2, flaky) == nullptr && table.size() == 2);
  Entry* c = table.get_or_create(2, flaky);
  CHECK(c && c->device == 2 && attempts == 2 && table.size() == 3);
  // many threads asking for the same new devices at once: exactly one creation each, everyone sees the same entry
  std::vector<std::thread> threads;
  std::vector<Entry*> seen(64, nullptr);
  const int before = created;
  for (int t = 0; t < 64; ++t) threads.emplace_back([&, t] { seen[t] = table.get_or_create(10 + t % 4, make); });
  for (auto& th : threads) th.join();
  CHECK(created == before + 4 && table.size() == 7);
  for (int t = 0; t < 64; ++t) CHECK(seen[t] && seen[t]->device == 10 + t % 4 && seen[t] == seen[t % 4]);
  int visited = 0, last = -1;
  table.for_each([&](int dev, Entry& e) { (void)e; if (dev > last) ++visited; last = dev; });
  CHECK(visited == 7);
  std::printf("device_table_test ok\n");
  return 0;
}
