"""per-launch FETCH_SIZE / WRITE_SIZE of the line kernels from a tools/r06_line_traffic.sh output directory"""
import csv, sys
O = sys.argv[1]
for fn, name in ((O + "/FETCH_SIZE_line1.csv", "FETCH_SIZE"), (O + "/WRITE_SIZE_line1.csv", "WRITE_SIZE")):
	for row in csv.DictReader(open(fn)):
		if row["Counter_Name"] == name and ("theta_line" in row["Kernel_Name"] or "ring_line" in row["Kernel_Name"]):
			k = "ring_line" if "ring_line" in row["Kernel_Name"] else ("theta_line to_cc" if "Li7ELi9E" in row["Kernel_Name"] else "theta_line from_cc_adjoint")
			print("%-10s %-28s %9.1f MB  %8.1f us" % (name, k, float(row["Counter_Value"])*1024/1e6, (int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))/1e3))
