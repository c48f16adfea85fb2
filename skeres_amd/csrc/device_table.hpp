// One entry per HIP device, created on first use under a lock and never destroyed before process exit.
// Holds what must exist once per DEVICE rather than once per process or per solver: the CU-masked queues of the
// look-ahead factorisation (chol_kernels.hip) — one process may drive several devices, one solver handle each
// (INTEGRATION.md section 4), and a queue created under device 0 cannot run kernels on device 1's memory.
// No HIP dependency: the creator is the caller's (tests/device_table_test.cpp exercises the logic on the CPU).
#pragma once
#include <map>
#include <memory>
#include <mutex>

namespace sk {

template <class T>
class PerDeviceTable {
 public:
  // The entry of `device`, created by create(device) -> T* (may return nullptr: nothing is stored, the next call
  // tries again).  At most one creation per device however many threads ask at once; entries keep their address.
  template <class Create>
  T* get_or_create(int device, Create create) {
    std::lock_guard<std::mutex> lock(mutex_);
    auto it = entries_.find(device);
    if (it != entries_.end()) return it->second.get();
    std::unique_ptr<T> fresh(create(device));
    if (!fresh) return nullptr;
    T* p = fresh.get();
    entries_[device] = std::move(fresh);
    return p;
  }
  T* find(int device) {
    std::lock_guard<std::mutex> lock(mutex_);
    auto it = entries_.find(device);
    return it == entries_.end() ? nullptr : it->second.get();
  }
  size_t size() {
    std::lock_guard<std::mutex> lock(mutex_);
    return entries_.size();
  }
  // visit every entry (process exit: release what the entries hold, in device order)
  template <class F>
  void for_each(F f) {
    std::lock_guard<std::mutex> lock(mutex_);
    for (auto& e : entries_) f(e.first, *e.second);
  }
  // serialises whole operations on an entry that must not interleave (the queue trial of one device)
  std::mutex& operation_mutex() { return op_mutex_; }

 private:
  std::mutex mutex_, op_mutex_;
  std::map<int, std::unique_ptr<T>> entries_;
};

}  // namespace sk
