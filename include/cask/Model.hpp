// Device model.  The reference's src/runtime/Model.hpp (:10-186) is an FPGA
// resource algebra (LUT/FF/BRAM/DSP of Max3/Max4/Max5 boards) used to REJECT
// and RANK design points without running them; on a GPU every point is
// measured, so what remains is the handful of constants a roofline needs.
// They are read from the HIP runtime (cask_hip_device_props_get), with the
// MI355X figures as documented fall-backs for reports written off-device.
#ifndef CASK_MODEL_HPP
#define CASK_MODEL_HPP

#include <sstream>
#include <string>

namespace cask {
namespace model {

struct HardwareModel {          // what one design point needs / achieves
  double memoryBandwidth = 0;   // GB/s, measured (algorithmic bytes / time)
  int ldsBytesPerWorkgroup = 0;
  int workgroups = 0;
  std::string to_string() const {
    std::stringstream s;
    s << ldsBytesPerWorkgroup << " " << workgroups << " " << memoryBandwidth;
    return s.str();
  }
};

class DeviceModel {
 public:
  virtual ~DeviceModel() {}
  virtual std::string getId() const = 0;
  virtual double hbmPeakGBs() const = 0;
  virtual int computeUnits() const = 0;
  virtual int ldsBytesPerCu() const = 0;
};

class Mi355xModel : public DeviceModel {
 public:
  std::string getId() const override { return "MI355X"; }
  double hbmPeakGBs() const override { return 8000.0; }   // HBM3E spec; ~6300 measured copy
  int computeUnits() const override { return 256; }
  int ldsBytesPerCu() const override { return 160 * 1024; }
};

inline std::ostream &operator<<(std::ostream &s, const DeviceModel &d) {
  s << "DeviceModel(" << d.getId() << ", " << d.hbmPeakGBs() << " GB/s, " << d.computeUnits() << " CUs)";
  return s;
}

}  // namespace model
}  // namespace cask

#endif  // CASK_MODEL_HPP
