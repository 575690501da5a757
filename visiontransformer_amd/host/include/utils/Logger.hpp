// Per-epoch CSV logger (reference: include/utils/Logger.hpp, src/utils/Logger.cpp): header
// epoch,epochs_total,train_loss,train_acc,test_loss,test_acc; 4 decimals; flushed per epoch.
#ifndef LOGGER_HPP
#define LOGGER_HPP

#include <fstream>
#include <iomanip>
#include <string>

class Logger {
public:
  Logger(const std::string &filename, bool overwrite = true);
  ~Logger();
  void log_epoch(int epoch, int epochs_total, float train_loss, float train_acc, float test_loss, float test_acc);

private:
  std::ofstream file;
  bool is_first_entry = true;
};

#endif  // LOGGER_HPP
